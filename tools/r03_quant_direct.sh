#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03r}; mkdir -p $O
for lib in lib/libkvq_hip.so lib/direct/libkvq_hip.so lib/libkvq_hip.so lib/direct/libkvq_hip.so; do
  line=$(KVQ_HIP_LIB=efficient-llm-inference_amd/$lib timeout -k 10 200 python bench.py --steps 6 --warmup 3 --no-subrecords --no-cpu-baseline 2>>$O/err.txt | tail -1)
  echo "$lib :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); q=j["roofline_quantise"]; print("int4", q["quant_int4"]["avg_launch_ms"], q["quant_int4"]["frac"], "int8", q["quant_int8"]["avg_launch_ms"], q["quant_int8"]["frac"])' 2>/dev/null || echo FAILED)" | tee -a $O/direct.txt
done
KVQ_HIP_LIB=efficient-llm-inference_amd/lib/direct/libkvq_hip.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=200 -k "oracle_quant_dequant or golden" 2>&1 | tail -2 | tee -a $O/direct.txt
