#!/bin/bash
# Quantise kernels: non-temporal vs write-back output stores x consecutive tiles per XCD (so that one XCD's L2 sees
# adjacent tiles' 256-byte output pieces and can merge them before they go to HBM). GPU box, repo root.
out=gpurun_out/r02af_quant_store_sweep.txt
: > $out
for rep in 1 2; do
for nts in 1 0; do for xg in 0 2 4 8 16 64; do
  timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --tunable quant_nt_stores=$nts --tunable quant_xcd_group=$xg > /tmp/q.json 2>/dev/null || exit 1
  python3 -c "
import json; j=json.load(open('/tmp/q.json')); q=j['roofline_quantise']
print('nt_stores=$nts xcd_group=%-3s  int4 %.4f ms (%.3f)  int8 %.4f ms (%.3f)' % ('$xg', q['quant_int4']['avg_launch_ms'], q['quant_int4']['frac'], q['quant_int8']['avg_launch_ms'], q['quant_int8']['frac']))" | tee -a $out
done; done; done
