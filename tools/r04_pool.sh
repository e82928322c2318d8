#!/bin/bash
# Round 4: what the chunk mean-pool's SEQUENTIAL summation order costs (make calib_pool: a blocked order, not the oracle's),
# and the gpt2-family quantise shapes with padded input rows.   -> gpurun_out/r04pool/
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04pool; mkdir -p $O
bash tools/sweep.sh r04pool --workload llama3_8b_evict_seq32k --steps 6 --warmup 2 --reps 3 -- "" "lib=pool_blocked" || exit 1
for S in gpt2m_int4_seq4k gpt2_shape_seq32k; do
  timeout -k 10 300 python3 bench.py --workload shape:$S --steps 24 > $O/shape_$S.json 2> $O/shape_$S.err || { tail $O/shape_$S.err; exit 1; }
  python3 -c "
import json,sys
j=json.loads(open('$O/shape_$S.json').read().strip().splitlines()[-1])
for k in ('int8','int4'):
    q=j['quant_'+k]; print('$S', k, 'contiguous', round(q['avg_launch_ms']*1e3,1), q['frac'], 'padded', round(q['padded_rows']['avg_launch_ms']*1e3,1), q['padded_rows']['frac'], 'dequant', j['dequant_'+k]['frac'])
"
done
