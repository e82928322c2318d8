#!/bin/bash
# ONE parameterised A-B sweep instead of a script per experiment (rounds 1-3 left 37 of those; tools/README.md lists the
# command line that replaces each). Runs `bench.py` once per variant and repetition and prints one line per run.
#
#   tools/sweep.sh <tag> [--workload W] [--reps N] [--steps S] [--warmup W] [--field F] [--extra "bench args"] -- VARIANT [VARIANT ...]
#
#   VARIANT   a quoted, space-separated list of KEY=VALUE tunables (kvq_set_tunable keys), optionally starting with
#             lib=<name>: <name> = default | ab | a directory under efficient-llm-inference_amd/lib/ holding a libkvq_hip.so
#             built by a `make calib_*` target (e.g. lib=fold_noacq, lib=prio1). "" = the shipped library, no tunables.
#             A variant with A-B keys and no lib= runs on the A-B library.
#   --field   which record of the line to print: roofline (default) | roofline_quantise | sharded | value
#   -> gpurun_out/<tag>/sweep.txt (+ one JSON line per run in gpurun_out/<tag>/runs.jsonl)
# Examples:
#   tools/sweep.sh r04fold --workload llama3_8b_decode_attn_seq16k_b8 --reps 2 -- "" "attn_fold=1" "lib=fold_noacq attn_fold=1" "lib=fold_nomerge attn_fold=1"
#   tools/sweep.sh r03prio --field roofline_quantise --extra "--no-subrecords --no-cpu-baseline" --steps 10 --reps 3 -- "" "lib=prio1" "lib=prio2"
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG=${1:?usage: tools/sweep.sh <tag> [options] -- VARIANT ...}; shift
W=llama3_8b_mixed_seq16k; REPS=1; STEPS=30; WARM=5; FIELD=roofline; EXTRA=""
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
  case "$1" in
    --workload) W=$2; shift 2;; --reps) REPS=$2; shift 2;; --steps) STEPS=$2; shift 2;; --warmup) WARM=$2; shift 2;;
    --field) FIELD=$2; shift 2;; --extra) EXTRA=$2; shift 2;; *) echo "unknown option $1" >&2; exit 2;;
  esac
done
[ "$1" = "--" ] && shift
[ $# -gt 0 ] || { echo "no variants" >&2; exit 2; }
O=gpurun_out/$TAG; mkdir -p $O
LIBDIR=$PWD/efficient-llm-inference_amd/lib
# the A-B keys, read from the library's own table (kvq_abi.hip): a variant that names one runs on the A-B library
AB_KEYS=$(python3 -c '
import re
src = open("efficient-llm-inference_amd/csrc/kvq_abi.hip").read()
print(" ".join(re.findall(r"\{\"([a-z0-9_]+)\", &Tunables::[a-z0-9_]+, true\}", src)))')
cat > $O/.fmt.py <<'PY'
import json, sys
field, rep, lib, variant = sys.argv[1:5]
try:
    j = json.loads(sys.stdin.read())
except Exception:
    print(f"rep {rep} [{lib}] {variant!r:44s} FAILED")
    sys.exit(1)


def one(r):
    return f"{r['avg_launch_ms'] * 1e3:9.2f} us  frac {r['frac']:.4f}  {r['kernel'][:70]}"


if field == "roofline":
    out = one(j["roofline"])
elif field == "roofline_quantise":
    out = "  |  ".join(f"{k} {one(v)}" for k, v in j["roofline_quantise"].items())
elif field == "sharded":
    out = f"{j['ms_per_step']} ms/step  {j['value']} GB/s  {j.get('kernels')}" + (f"  two_phase {j['two_phase']['ms_per_step']} ms" if j.get("two_phase") else "")
else:
    out = f"{j['value']} {j['unit']}  {j['ms_per_step']} ms/step"
print(f"rep {rep} [{lib}] {variant!r:44s} {out}")
PY
failed=0
for rep in $(seq 1 $REPS); do
  for V in "$@"; do
    lib=""; args=""; needs_ab=0
    for kv in $V; do
      case "$kv" in
        lib=*) lib=${kv#lib=};;
        *) args="$args --tunable $kv"; for k in $AB_KEYS; do [ "${kv%%=*}" = "$k" ] && needs_ab=1; done;;
      esac
    done
    if [ -z "$lib" ]; then if [ $needs_ab = 1 ]; then lib=ab; else lib=default; fi; fi
    if [ "$lib" = default ]; then path=$LIBDIR/libkvq_hip.so; else path=$LIBDIR/$lib/libkvq_hip.so; fi
    if [ ! -f "$path" ]; then echo "missing $path (make -C efficient-llm-inference_amd/csrc ab | calib_*)" | tee -a $O/sweep.txt; failed=$((failed+1)); continue; fi
    line=$(KVQ_HIP_LIB=$path timeout -k 10 400 python3 bench.py --workload $W --steps $STEPS --warmup $WARM $EXTRA $args 2>>$O/sweep.err | tail -1)
    echo "$line" >> $O/runs.jsonl
    echo "$line" | python3 $O/.fmt.py "$FIELD" "$rep" "$lib" "$V" | tee -a $O/sweep.txt || failed=$((failed+1))
  done
done
rm -f $O/.fmt.py
exit $failed
