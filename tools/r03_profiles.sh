#!/bin/bash
# Evidence run of round 3: rocprofv3 kernel statistics of the default bench line (every sub-record's kernels), of the
# batch-8 decode attention and of the batch-sharded quantise; PMC passes (traffic, TCC, TA / TCP, SQ) of the headline.
# PART=stats | pmc runs one half (each fits one 20-minute gpurun call).
set -o pipefail
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/${1:-r03p}; mkdir -p $O; export TMPDIR=/tmp
stats() {  # name, bench args...
  local name=$1; shift
  echo "== stats $name" | tee -a $O/progress.txt
  (cd /tmp && timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py "$@" > $O/$name.json 2> $O/$name.err) || echo "stats $name FAILED" | tee -a $O/progress.txt
  find $O/$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${name}_kernel_stats.csv
  rm -rf $O/$name   # the raw trace is tens of MiB; the statistics are what is kept
  grep "kvq::" $O/${name}_kernel_stats.csv | cut -c1-160 | head -12
}
if [ "${PART:-all}" != "pmc" ]; then
stats headline --steps 20 --warmup 5 --no-subrecords --no-cpu-baseline
stats evict --steps 6 --warmup 2 --workload llama3_8b_evict_seq32k
stats shape_gpt2m --steps 24 --warmup 2 --workload shape:gpt2m_int4_seq4k
stats shape_gpt2 --steps 24 --warmup 2 --workload shape:gpt2_shape_seq32k
stats decode_gpt2 --steps 1 --warmup 1 --workload decode:gpt2:quant_int8:512:512
stats attn_b8 --steps 30 --warmup 5 --workload llama3_8b_decode_attn_seq16k_b8
stats attn_b1 --steps 30 --warmup 5 --workload llama3_8b_decode_attn_seq16k
stats shardq --steps 20 --warmup 5 --workload llama3_8b_batch64_sharded_prefill512
echo "== plain runs" | tee -a $O/progress.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --workload llama3_8b_batch64_sharded_prefill512 > $O/bench_shardq.json 2>> $O/plain.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --workload llama3_8b_batch64_sharded_append > $O/bench_shardq_append.json 2>> $O/plain.err
fi
if [ "${PART:-all}" != "stats" ]; then
echo "== pmc headline" | tee -a $O/progress.txt
bash tools/r03_pmc.sh $(basename $O)/pmc_headline --steps 4 --warmup 2 --no-cpu-baseline --no-subrecords; echo "pmc headline rc=$?" | tee -a $O/progress.txt
echo "== pmc traffic attn b8 / shardq / evict" | tee -a $O/progress.txt
PMC_ONLY_TRAFFIC=1 bash tools/r03_pmc.sh $(basename $O)/pmc_attn_b8 --steps 4 --warmup 2 --workload llama3_8b_decode_attn_seq16k_b8; echo "rc=$?" | tee -a $O/progress.txt
PMC_ONLY_TRAFFIC=1 bash tools/r03_pmc.sh $(basename $O)/pmc_shardq --steps 4 --warmup 2 --workload llama3_8b_batch64_sharded_prefill512; echo "rc=$?" | tee -a $O/progress.txt
PMC_ONLY_TRAFFIC=1 bash tools/r03_pmc.sh $(basename $O)/pmc_evict --steps 3 --warmup 1 --workload llama3_8b_evict_seq32k; echo "rc=$?" | tee -a $O/progress.txt
for d in pmc_headline pmc_attn_b8 pmc_shardq pmc_evict; do rm -rf $O/$d/p[0-9]*/; done   # keep summary.csv / traffic.csv / logs
fi
[ -f $O/bench_shardq.json ] && cut -c1-400 $O/bench_shardq.json; du -sh $O
