#!/bin/bash
# rocprofv3 kernel statistics of the batch-sharded decode append (two_phase sub-record = the phases a rank of an N > 1 job runs),
# with the few-token abs-max kernel (default) and with the tile walk's atomics (quant_few_tokens=0)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD; O=$R/gpurun_out/r04z; mkdir -p $O; export TMPDIR=/tmp
for V in 1 0; do
  name=shardq_append_few_tokens_$V
  (cd /tmp && timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py --steps 200 --warmup 20 --workload llama3_8b_batch64_sharded_append --tunable quant_few_tokens=$V > $O/${name}_under_rocprof.json 2> $O/$name.err) || { tail -5 $O/$name.err; exit 1; }
  find $O/$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${name}_kernel_stats.csv
  rm -rf $O/$name
  echo "== quant_few_tokens=$V"; grep "kvq::" $O/${name}_kernel_stats.csv | cut -c1-150 | head -6
done
