#!/bin/bash
# merge of <= 16 splits by one wave per head (attn_merge_wave=1, shipped if it wins) against the workgroup merge
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03mw}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_attn.py tests/test_gpu_benchmarker.py -m gpu -x -q --timeout=300 > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee $O/progress.txt; tail -4 $O/pytest.txt | tee -a $O/progress.txt
run() {
  local w=$1; shift
  local args=""; for kv in "$@"; do args="$args --tunable $kv"; done
  local line
  line=$(timeout -k 10 300 python bench.py --steps 30 --warmup 5 --workload $w $args 2>>$O/sweep.err | tail -1)
  echo "$w $* :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(round(r["avg_launch_ms"]*1e3,2), "us/layer", r["frac"], r["kernel"][:90])' 2>/dev/null || echo FAILED)" | tee -a $O/sweep.txt
}
for rep in 1 2 3; do
  run llama3_8b_decode_attn_seq16k_b8 attn_merge_wave=1
  run llama3_8b_decode_attn_seq16k_b8 attn_merge_wave=0
done
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/attn_b8 -- python3 $OLDPWD/bench.py --steps 30 --warmup 5 --workload llama3_8b_decode_attn_seq16k_b8 > $OLDPWD/$O/attn_b8.json 2> $OLDPWD/$O/attn_b8.err)
find $O/attn_b8 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/attn_b8_kernel_stats.csv; rm -rf $O/attn_b8
grep "kvq::" $O/attn_b8_kernel_stats.csv | cut -c1-170 | head -3
