#!/bin/bash
# the headline step three times on whatever box the call lands on: box-to-box and run-to-run spread of value / roofline.frac
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03rep}; mkdir -p $O
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-subrecords --no-cpu-baseline 2>>$O/err.txt | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; q=j['roofline_quantise']
print('value', j['value'], 'int4', r['frac'], round(r['avg_launch_ms']*1e3,1), 'int8', j['roofline_k']['frac'], 'quant', q['quant_int4']['frac'], q['quant_int8']['frac'])" | tee -a $O/repeat.txt
done
