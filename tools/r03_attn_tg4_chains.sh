#!/bin/bash
# TG = 4 score product: one accumulator per plane (chains of 8 dependent MFMAs, shipped) against two (chains of 4; make calib_tg4)
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03tgc}; mkdir -p $O
L=efficient-llm-inference_amd/lib
KVQ_HIP_LIB=$L/tg4c2/libkvq_hip.so timeout -k 10 600 python -m pytest tests/test_gpu_attn.py -m gpu -x -q --timeout=300 -k "lds_staged or full_size" > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee $O/progress.txt; tail -3 $O/pytest.txt
for rep in 1 2 3; do for v in shipped tg4c2; do
  lib=$L/$v/libkvq_hip.so; [ $v = shipped ] && lib=$L/libkvq_hip.so
  line=$(KVQ_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 30 --warmup 5 --workload llama3_8b_decode_attn_seq16k_b8 2>>$O/err.txt | tail -1)
  echo "$v :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(round(r["avg_launch_ms"]*1e3,2), r["frac"])')" | tee -a $O/sweep.txt
done; done
