#!/bin/bash
# Cache-policy bits of the quantise tile kernel (make calib_aux) on the headline shape + pytest durations of the shipped suite.
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03f}; mkdir -p $O
for c in shipped 2_0 2_3 2_18 2_19 2_17 0_2 3_2 18_2 shipped; do
  lib=efficient-llm-inference_amd/lib/aux_$c/libkvq_hip.so; [ $c = shipped ] && lib=efficient-llm-inference_amd/lib/libkvq_hip.so
  line=$(KVQ_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 6 --warmup 3 --no-subrecords --no-cpu-baseline 2>>$O/err.txt | tail -1)
  echo "ld_st=$c :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); q=j["roofline_quantise"]; print("int4", q["quant_int4"]["avg_launch_ms"], q["quant_int4"]["frac"], "int8", q["quant_int8"]["avg_launch_ms"], q["quant_int8"]["frac"])' 2>/dev/null || echo FAILED)" | tee -a $O/aux.txt
done
echo "== pytest durations" | tee -a $O/progress.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout=300 --durations=40 > $O/pytest_durations.txt 2>&1; echo "rc=$?" | tee -a $O/progress.txt
grep -A45 "slowest" $O/pytest_durations.txt | head -60
