#!/bin/bash
# Evidence run of round 4 on the final binary: bench lines (default + every secondary workload), rocprofv3 kernel statistics per
# workload, PMC traffic passes of the kernels that are new this round (gather_rows_k, the head_dim-64 ring kernel), test logs.
# PART=lines | stats | pmc runs one part (each fits one gpurun call).   -> gpurun_out/<tag>/
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD; O=$R/gpurun_out/${1:-r04y}; mkdir -p $O; export TMPDIR=/tmp
stats() {  # name, bench args...
  local name=$1; shift
  echo "== stats $name" | tee -a $O/progress.txt
  (cd /tmp && timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py "$@" > $O/${name}_under_rocprof.json 2> $O/$name.err) || echo "stats $name FAILED" | tee -a $O/progress.txt
  find $O/$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${name}_kernel_stats.csv
  rm -rf $O/$name   # the raw trace is tens of MiB; the statistics are what is kept
  grep "kvq::" $O/${name}_kernel_stats.csv | cut -c1-170 | head -12
}
if [ "${PART:-all}" = "all" ] || [ "$PART" = "lines" ]; then
  echo "== tests" | tee -a $O/progress.txt
  timeout -k 10 900 python3 -m pytest tests -q -m gpu --durations=8 > $O/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?" | tee -a $O/progress.txt; tail -3 $O/pytest_gpu.txt
  timeout -k 10 300 python3 -m pytest tests/test_bench_cli.py -q -m benchcli > $O/pytest_benchcli.txt 2>&1; echo "pytest benchcli rc=$?" | tee -a $O/progress.txt
  timeout -k 10 1100 python3 -m pytest tests -q -m ab > $O/pytest_ab.txt 2>&1; echo "pytest ab rc=$?" | tee -a $O/progress.txt; tail -3 $O/pytest_ab.txt
  echo "== default line" | tee -a $O/progress.txt
  timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?" | tee -a $O/progress.txt
  for W in llama3_8b_decode_attn_seq16k llama3_8b_decode_attn_seq16k_b8 gpt2_decode_attn_seq1k llama2_7b_decode_attn_seq4k_b8 llama32_1b_decode_attn_seq16k_b8; do
    timeout -k 10 300 python3 bench.py --workload $W --steps 30 --warmup 5 > $O/bench_$W.json 2>> $O/plain.err; echo "$W rc=$?" | tee -a $O/progress.txt
  done
  timeout -k 10 600 python3 bench.py --workload llama3_8b_sparse_seq32k > $O/bench_sparse_seq32k.json 2>> $O/plain.err; echo "sparse rc=$?" | tee -a $O/progress.txt
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --workload llama3_8b_batch64_sharded_prefill512 > $O/bench_sharded_quant_prefill512.json 2>> $O/plain.err
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --workload llama3_8b_batch64_sharded_append > $O/bench_sharded_quant_append.json 2>> $O/plain.err
fi
if [ "${PART:-all}" = "all" ] || [ "$PART" = "stats" ]; then
  stats headline --steps 20 --warmup 5 --no-subrecords --no-cpu-baseline
  stats evict_sparse --workload llama3_8b_sparse_seq32k
  stats evict --steps 6 --warmup 2 --workload llama3_8b_evict_seq32k
  stats attn_b8 --steps 30 --warmup 5 --workload llama3_8b_decode_attn_seq16k_b8
  stats attn_b1 --steps 30 --warmup 5 --workload llama3_8b_decode_attn_seq16k
  stats attn_llama32_1b_b8 --steps 30 --warmup 5 --workload llama32_1b_decode_attn_seq16k_b8
  stats attn_llama2_7b_b8 --steps 30 --warmup 5 --workload llama2_7b_decode_attn_seq4k_b8
  stats shape_gpt2m --steps 24 --warmup 2 --workload shape:gpt2m_int4_seq4k
  stats shape_gpt2 --steps 24 --warmup 2 --workload shape:gpt2_shape_seq32k
  stats shardq_prefill512 --steps 20 --warmup 5 --workload llama3_8b_batch64_sharded_prefill512
  stats shardq_append --steps 200 --warmup 20 --workload llama3_8b_batch64_sharded_append
fi
if [ "${PART:-all}" = "all" ] || [ "$PART" = "pmc" ]; then
  echo "== pmc traffic" | tee -a $O/progress.txt
  PMC_ONLY_TRAFFIC=1 bash tools/r03_pmc.sh $(basename $O)/pmc_sparse --workload llama3_8b_sparse_seq32k; echo "rc=$?" | tee -a $O/progress.txt
  PMC_ONLY_TRAFFIC=1 bash tools/r03_pmc.sh $(basename $O)/pmc_attn_llama32_1b_b8 --steps 4 --warmup 2 --workload llama32_1b_decode_attn_seq16k_b8; echo "rc=$?" | tee -a $O/progress.txt
  PMC_ONLY_TRAFFIC=1 bash tools/r03_pmc.sh $(basename $O)/pmc_headline --steps 4 --warmup 2 --no-cpu-baseline --no-subrecords; echo "rc=$?" | tee -a $O/progress.txt
  for d in pmc_sparse pmc_attn_llama32_1b_b8 pmc_headline; do rm -rf $O/$d/p[0-9]*/; done
fi
du -sh $O
