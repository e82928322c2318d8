#!/bin/bash
# Round 4 whole-tree check: pytest -m gpu (shipped library), the round's A-B tests (A-B library), the N3 workload with both
# gather kernels, the default bench line.   -> gpurun_out/r04check/
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD; O=$R/gpurun_out/r04check; mkdir -p $O; export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $O/progress.txt
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu --durations=15 > $O/pytest_gpu.txt 2>&1 || { tail -40 $O/pytest_gpu.txt; exit 1; }
tail -22 $O/pytest_gpu.txt
echo "== pytest -m benchcli" | tee -a $O/progress.txt
timeout -k 10 600 python3 -m pytest tests/test_bench_cli.py -x -q -m benchcli > $O/pytest_benchcli.txt 2>&1 || { tail -40 $O/pytest_benchcli.txt; exit 1; }
tail -2 $O/pytest_benchcli.txt
echo "== pytest -m ab -k merge_inside" | tee -a $O/progress.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_attn.py -x -q -m ab -k "merge_inside" > $O/pytest_ab_fold.txt 2>&1 || { tail -40 $O/pytest_ab_fold.txt; exit 1; }
tail -2 $O/pytest_ab_fold.txt
for V in 1 0; do
  echo "== sparse gather_rows=$V" | tee -a $O/progress.txt
  timeout -k 10 600 python3 bench.py --workload llama3_8b_sparse_seq32k --tunable gather_rows=$V > $O/sparse_gather_rows_$V.json 2> $O/sparse_gather_rows_$V.err || { tail -20 $O/sparse_gather_rows_$V.err; exit 1; }
done
echo "== default bench line" | tee -a $O/progress.txt
timeout -k 10 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python3 - "$O" <<'PY'
import json, os, sys
O = sys.argv[1]
for V in (1, 0):
    j = json.loads(open(os.path.join(O, f"sparse_gather_rows_{V}.json")).read().strip().splitlines()[-1])
    for k in ("trim_kv_strided", "trim_kv_block_old", "trim_kv_budget_old", "trim_kv_prefix_window", "paged_get_kv"):
        r = j[k]
        print(f"gather_rows={V} {k:24s} {r['avg_launch_ms'] * 1e3:9.1f} us  {r['achieved']:8.1f} GB/s  frac {r['frac']:.4f}  {r['kernel'][:60]}  kept {r.get('kept_tokens')}")
j = json.loads(open(os.path.join(O, "bench_default.json")).read().strip().splitlines()[-1])
print("default:", j["value"], j["roofline"]["frac"], j["run_s"], j["phases_s"])
print("quantise:", {k: (v["frac"], v["padded_rows"]["frac"], v["outlier_channels"]["frac"]) for k, v in j["roofline_quantise"].items()})
print("cpu:", j["cpu_baseline"]["host_cpu"], j["cpu_baseline"]["timer_quantum_suspects"])
print("sparse in default:", {k: v.get("frac") for k, v in j["configs"].get("llama3_8b_sparse_seq32k", {}).items() if isinstance(v, dict) and "frac" in v})
PY
echo done | tee -a $O/progress.txt
