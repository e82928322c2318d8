#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03o}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_attn.py tests/test_gpu_benchmarker.py -m gpu -x -q --timeout=300 > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee $O/progress.txt; tail -3 $O/pytest.txt
timeout -k 10 600 python -m pytest tests/test_gpu_attn.py -m ab -x -q --timeout=300 -k "merge" > $O/pytest_ab.txt 2>&1; echo "pytest ab rc=$?" | tee -a $O/progress.txt; tail -3 $O/pytest_ab.txt
for w in llama3_8b_decode_attn_seq16k_b8 llama3_8b_decode_attn_seq16k llama3_8b_decode_attn_seq16k_b8 llama3_8b_decode_attn_seq16k; do
  line=$(timeout -k 10 200 python bench.py --steps 30 --warmup 5 --workload $w 2>>$O/err.txt | tail -1)
  echo "$w :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(round(r["avg_launch_ms"]*1e3,2), "us/layer", r["frac"], r["kernel"][:90])')" | tee -a $O/attn.txt
done
