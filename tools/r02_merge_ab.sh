#!/bin/bash
# A/B of the merge kernels (attn_merge_fast = 1 requests all operands up front, 0 = the chained one) on the
# decode-attention workloads + rocprofv3 kernel tables. GPU box, repo root.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r02x_merge_ab.jsonl
: > $out
export TMPDIR=/tmp
for wl in llama3_8b_decode_attn_seq16k llama3_8b_decode_attn_seq16k_b8 llama32_1b_decode_attn_seq16k_b8; do
  for fast in 1 0 1 0; do
    echo "## $wl attn_merge_fast=$fast" >> $out
    timeout -k 10 300 python3 bench.py --workload $wl --steps 20 --warmup 3 --tunable attn_merge_fast=$fast >> $out 2>> $R/gpurun_out/r02x_merge_ab.err || exit 1
  done
done
for fast in 1 0; do
  for wl in llama3_8b_decode_attn_seq16k llama3_8b_decode_attn_seq16k_b8; do
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mab_${wl}_$fast -- python3 $R/bench.py --workload $wl --steps 10 --warmup 2 --tunable attn_merge_fast=$fast > /dev/null 2>&1) || exit 1
    f=$(find /tmp/mab_${wl}_$fast -name '*kernel_stats.csv' | head -1)
    head -6 "$f" > $R/gpurun_out/r02x_kernel_stats_${wl}_mergefast$fast.csv
  done
done
python3 - <<'PY'
import json, os
for ln in open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r02x_merge_ab.jsonl"):
    if ln.startswith("##"):
        tag = ln.strip()
    elif ln.startswith("{"):
        j = json.loads(ln)
        print(tag, "ms_per_step", j["ms_per_step"], "value", j["value"], j.get("roofline", {}).get("avg_launch_ms"))
PY
