#!/bin/bash
# Round 4: where the in-launch merge's time goes — the shipped fold (agent acquire), no acquire, no merge at all (calibration
# builds, `make -C efficient-llm-inference_amd/csrc calib_fold`), against two launches; batch 8, 16 K tokens  -> gpurun_out/r04fold2/
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD; O=$R/gpurun_out/r04fold2; mkdir -p $O; export TMPDIR=/tmp
L=$R/efficient-llm-inference_amd/lib
timeout -k 10 600 python3 -m pytest tests/test_gpu_attn.py -x -q -m gpu -k "merge_inside" > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
KVQ_HIP_LIB=$L/fold_noacq/libkvq_hip.so timeout -k 10 600 python3 -m pytest tests/test_gpu_attn.py -x -q -m gpu -k "merge_inside" > $O/pytest_noacq.txt 2>&1 || { tail -30 $O/pytest_noacq.txt; exit 1; }
tail -2 $O/pytest.txt $O/pytest_noacq.txt
W=llama3_8b_decode_attn_seq16k_b8
for rep in 1 2; do
  for V in two_launches fold fold_noacq fold_nomerge; do
    T=""; LIB=$L/libkvq_hip.so
    [ $V = two_launches ] && T="--tunable attn_fold=0"
    [ $V = fold_noacq ] && LIB=$L/fold_noacq/libkvq_hip.so
    [ $V = fold_nomerge ] && LIB=$L/fold_nomerge/libkvq_hip.so
    echo "== $V rep $rep" | tee -a $O/progress.txt
    KVQ_HIP_LIB=$LIB timeout -k 10 300 python3 bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline $T > $O/${V}_$rep.json 2> $O/${V}_$rep.err || exit 1
  done
done
python3 - "$O" <<'PY'
import json, glob, os, sys
O = sys.argv[1]
for f in sorted(glob.glob(os.path.join(O, "*.json"))):
    j = json.loads(open(f).read().strip().splitlines()[-1]); r = j.get("roofline", {})
    print(f"{os.path.basename(f):30s} per-layer us {1e3 * r.get('avg_launch_ms', 0):.2f}  frac {r.get('frac')}")
PY
