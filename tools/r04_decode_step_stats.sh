#!/bin/bash
# rocprofv3 kernel statistics of kvq_decode_step (attention + fused new-token append) and kvq_decode_attn at batch 64 / 16
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD; O=$R/gpurun_out/r04ac; mkdir -p $O; export TMPDIR=/tmp
for B in 64 16; do for W in decode_step decode_attn; do
  name=b${B}_$W
  (cd /tmp && KVQ_SWEEP_ONLY=$B:$W timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/tools/decode_step_batch_sweep.py > $O/$name.json 2> $O/$name.err) || { tail -5 $O/$name.err; exit 1; }
  find $O/$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${name}_kernel_stats.csv
  rm -rf $O/$name
  echo "== $name"; grep "kvq::" $O/${name}_kernel_stats.csv | cut -c1-150 | head -4
done; done
