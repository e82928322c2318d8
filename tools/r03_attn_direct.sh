#!/bin/bash
# ring kernel, TG = 4 path with the serial chain cut down (consume_direct): tests, then the batch-8 / other workloads
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03dir}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_attn.py tests/test_gpu_benchmarker.py -m gpu -x -q --timeout=300 > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee $O/progress.txt; tail -4 $O/pytest.txt | tee -a $O/progress.txt
for rep in 1 2 3; do
  line=$(timeout -k 10 300 python bench.py --steps 30 --warmup 5 --workload llama3_8b_decode_attn_seq16k_b8 2>>$O/err.txt | tail -1)
  echo "b8 :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(round(r["avg_launch_ms"]*1e3,2), r["frac"], r["kernel"][:60])')" | tee -a $O/sweep.txt
done
