#!/bin/bash
# Quantise variants re-measured now that the kernels really issue non-temporal accesses (they did not before: a
# run-time `if (flag) nt_load else load` is merged by the compiler). GPU box, repo root.
out=gpurun_out/r02av_quant_variants_with_nt.txt
: > $out
run() { timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > /tmp/q.json 2>/dev/null || { echo "FAILED $*" | tee -a $out; return; }
  python3 -c "
import json,sys; j=json.load(open('/tmp/q.json')); q=j['roofline_quantise']
print('%-60s int4 %.4f ms (%.3f)  int8 %.4f ms (%.3f)' % (' '.join(sys.argv[1:]), q['quant_int4']['avg_launch_ms'], q['quant_int4']['frac'], q['quant_int8']['avg_launch_ms'], q['quant_int8']['frac']))" "$@" | tee -a $out; }
for rep in 1 2; do
  run
  run --tunable quant_geo128=0
  run --tunable quant_tpw=2
  run --tunable quant_tpw=4
  run --tunable quant_nv=4
  run --tunable quant_nv=16
  run --tunable quant_block=128
  run --tunable quant_block=256
  run --tunable quant_xcd_group=8
  run --tunable quant_nt_stores=1
  run --tunable quant_nt_stores=0
done
