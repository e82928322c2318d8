#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03g}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_attn.py tests/test_sharded_batch.py -m gpu -x -q --timeout=300 > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee $O/progress.txt; tail -3 $O/pytest.txt
for cb in 0 67108864 268435456; do
  line=$(KVQ_SHARD_CHUNK_BYTES=$cb timeout -k 10 200 python bench.py --steps 20 --warmup 5 --workload llama3_8b_batch64_sharded_prefill512 2>>$O/err.txt | tail -1)
  echo "chunk_bytes=$cb :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "GB/s", j["ms_per_step"], "ms chunks", j["config"]["layer_chunks"])')" | tee -a $O/shardq.txt
done
for w in llama3_8b_decode_attn_seq16k_b8 llama3_8b_decode_attn_seq16k; do
  line=$(timeout -k 10 200 python bench.py --steps 30 --warmup 5 --workload $w 2>>$O/err.txt | tail -1)
  echo "$w :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(round(r["avg_launch_ms"]*1e3,2), "us/layer", r["frac"], r["kernel"][:70])')" | tee -a $O/attn.txt
done
