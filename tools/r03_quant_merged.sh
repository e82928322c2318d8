#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03m}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m ab -x -q --timeout=300 -k "tile_sizes_and_instantiations" > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee $O/progress.txt; tail -3 $O/pytest.txt | tee -a $O/progress.txt
AB=efficient-llm-inference_amd/lib/ab/libkvq_hip.so
for t in "quant_tile_tpw=0" "quant_tile_tpw=2" "quant_tile_tpw=4" "quant_tile_tpw=0" "quant_tile_tpw=2"; do
  line=$(KVQ_HIP_LIB=$AB timeout -k 10 200 python bench.py --steps 6 --warmup 3 --no-subrecords --no-cpu-baseline --tunable $t 2>>$O/err.txt | tail -1)
  echo "$t :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); q=j["roofline_quantise"]; print("int4", q["quant_int4"]["avg_launch_ms"], q["quant_int4"]["frac"], "int8", q["quant_int8"]["avg_launch_ms"], q["quant_int8"]["frac"], q["quant_int4"]["kernel"][:44])' 2>/dev/null || echo FAILED)" | tee -a $O/merged.txt
done
