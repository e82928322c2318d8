#!/bin/bash
# attention ring kernel with ONE score output per tile (TG = 4, <= 4 query heads per kv head) against one per 16-token group
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03tg}; mkdir -p $O
echo "== pytest" | tee $O/progress.txt
timeout -k 10 900 python -m pytest tests/test_gpu_attn.py tests/test_gpu_benchmarker.py -m gpu -x -q --timeout=300 > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/progress.txt; tail -4 $O/pytest.txt | tee -a $O/progress.txt
timeout -k 10 900 python -m pytest tests/test_gpu_attn.py -m ab -x -q --timeout=300 -k "lds" > $O/pytest_ab.txt 2>&1
echo "pytest ab rc=$?" | tee -a $O/progress.txt; tail -4 $O/pytest_ab.txt | tee -a $O/progress.txt
AB=efficient-llm-inference_amd/lib/ab/libkvq_hip.so
run() {
  local w=$1; shift
  local args=""; for kv in "$@"; do args="$args --tunable $kv"; done
  local line
  line=$(KVQ_HIP_LIB=$AB timeout -k 10 300 python bench.py --steps 30 --warmup 5 --workload $w $args 2>>$O/sweep.err | tail -1)
  echo "$w $* :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(round(r["avg_launch_ms"]*1e3,2), "us/layer", r["frac"], r["kernel"][:70])' 2>/dev/null || echo FAILED)" | tee -a $O/sweep.txt
}
for rep in 1 2; do
  run llama3_8b_decode_attn_seq16k_b8 attn_tg=0
  run llama3_8b_decode_attn_seq16k_b8 attn_tg=1
done
for w in llama3_8b_decode_attn_seq16k_b8 llama3_8b_decode_attn_seq16k llama2_7b_decode_attn_seq4k_b8 llama32_1b_decode_attn_seq16k_b8 gpt2_decode_attn_seq1k; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --workload $w > $O/bench_$w.json 2>> $O/bench_attn.err; echo "$w rc=$?" | tee -a $O/progress.txt
done
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/attn_b8 -- python3 $OLDPWD/bench.py --steps 30 --warmup 5 --workload llama3_8b_decode_attn_seq16k_b8 > $OLDPWD/$O/attn_b8.json 2> $OLDPWD/$O/attn_b8.err)
find $O/attn_b8 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/attn_b8_kernel_stats.csv; rm -rf $O/attn_b8
grep "kvq::" $O/attn_b8_kernel_stats.csv | cut -c1-170 | head -4
PMC_ONLY_TRAFFIC=1 bash tools/r03_pmc.sh $(basename $O)/pmc_attn_b8 --steps 4 --warmup 2 --workload llama3_8b_decode_attn_seq16k_b8; echo "pmc rc=$?" | tee -a $O/progress.txt
rm -rf $O/pmc_attn_b8/p[0-9]*/
