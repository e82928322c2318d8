#!/bin/bash
# after the last attention change: rocprofv3 kernel statistics of the batch-8 / batch-1 attention workloads and the batch-8 PMC traffic passes
set -o pipefail
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/${1:-r03zz}; mkdir -p $O; export TMPDIR=/tmp
for pair in "attn_b8 llama3_8b_decode_attn_seq16k_b8" "attn_b1 llama3_8b_decode_attn_seq16k"; do
  set -- $pair
  (cd /tmp && timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -- python3 $R/bench.py --steps 30 --warmup 5 --workload $2 > $O/$1.json 2> $O/$1.err) || echo "stats $1 FAILED" | tee -a $O/progress.txt
  find $O/$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/$1_kernel_stats.csv; rm -rf $O/$1
  grep "kvq::" $O/$1_kernel_stats.csv | cut -c1-170 | head -3
done
PMC_ONLY_TRAFFIC=1 bash tools/r03_pmc.sh $(basename $O)/pmc_attn_b8 --steps 4 --warmup 2 --workload llama3_8b_decode_attn_seq16k_b8; echo "pmc rc=$?" | tee -a $O/progress.txt
rm -rf $O/pmc_attn_b8/p[0-9]*/
