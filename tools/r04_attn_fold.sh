#!/bin/bash
# Round 4, VERDICT item 2: the attention merge inside the partial launch (attn_fold) — tests, then A-B timings of the batch-8
# workload (fold in kvq_decode_step_layers = default, attn_fold=0 = two launches, per-layer calls with and without a memset + fold)
#   -> gpurun_out/r04fold/
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD; O=$R/gpurun_out/r04fold; mkdir -p $O; export TMPDIR=/tmp
echo "== tests" | tee $O/progress.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_attn.py -x -q -m gpu -k "merge_inside or lds_staged or step_layers or merge_by_one_wave or full_size" > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
W=llama3_8b_decode_attn_seq16k_b8
for rep in 1 2 3; do
  for V in "default" "attn_fold=0"; do
    echo "== $W $V rep $rep" | tee -a $O/progress.txt
    if [ "$V" = default ]; then T=""; else T="--tunable $V"; fi
    timeout -k 10 300 python3 bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline $T > $O/b8_${V/=/_}_$rep.json 2> $O/b8_${V/=/_}_$rep.err || exit 1
  done
done
for V in "attn_fold=1" "attn_fold=0"; do
  echo "== $W per-layer calls $V" | tee -a $O/progress.txt
  timeout -k 10 300 python3 bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --per-layer-calls --tunable $V > $O/b8_perlayer_${V/=/_}.json 2> $O/b8_perlayer_${V/=/_}.err || exit 1
done
for W2 in llama3_8b_decode_attn_seq16k llama32_1b_decode_attn_seq16k_b8 llama2_7b_decode_attn_seq4k_b8; do
  echo "== $W2" | tee -a $O/progress.txt
  timeout -k 10 300 python3 bench.py --workload $W2 --steps 30 --warmup 5 --no-cpu-baseline > $O/${W2}.json 2> $O/${W2}.err || exit 1
done
echo "== rocprof b8 default" | tee -a $O/progress.txt
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b8 -- python3 $R/bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline > $O/prof_b8.json 2> $O/prof_b8.err) || exit 1
find $O/prof_b8 -name "*kernel_stats.csv" -exec cp {} $O/prof_b8_kernel_stats.csv \;
find $O/prof_b8 -name "*.csv" -size +1M -delete
echo "== default bench line" | tee -a $O/progress.txt
cat /proc/self/cgroup > $O/cgroup.txt 2>&1; cat /sys/fs/cgroup/cpu.max >> $O/cgroup.txt 2>&1; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us >> $O/cgroup.txt 2>&1; nproc >> $O/cgroup.txt
timeout -k 10 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
python3 - "$O" <<'PY'
import json, glob, os, sys
O = sys.argv[1]
for f in sorted(glob.glob(os.path.join(O, "*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    r = j.get("roofline", {})
    print(f"{os.path.basename(f):50s} ms/step {j.get('ms_per_step')}  per-layer us {1e3 * r.get('avg_launch_ms', 0):.2f}  frac {r.get('frac')}  kernels {str(r.get('kernel'))[:90]}")
PY
echo done | tee -a $O/progress.txt
