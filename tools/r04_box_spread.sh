#!/bin/bash
# One default bench line (no CPU baseline) on whatever box this call lands on: the pool / window / headline fractions beside the
# card's state under load. Run it in several gpurun calls (each gets a fresh box) -> gpurun_out/r04v/box_<time>.json
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04v; mkdir -p $O
F=$O/box_$(date +%H%M%S).json
timeout -k 10 500 python3 bench.py --no-cpu-baseline > $F 2>> $O/err.txt || exit 1
python3 - "$F" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ev = j["configs"]["llama3_8b_evict_seq32k"]
ds, dp = j["device_state"], ev.get("device_state", {})
print(json.dumps({"headline_int4": j["roofline"]["frac"], "int8": j["roofline_k"]["frac"], "pool": ev["roofline"]["frac"], "pool_ms": ev["roofline"]["avg_launch_ms"],
                  "window": ev["roofline_window"]["frac"], "quant": {k: v["frac"] for k, v in j["roofline_quantise"].items()},
                  "under_headline": {k: ds.get(k) for k in ("sclk_mhz", "mclk_mhz", "fclk_mhz", "Current Socket Graphics Package Power (W)", "Temperature (Sensor junction) (C)", "Temperature (Sensor memory) (C)")},
                  "under_pool": {k: dp.get(k) for k in ("sclk_mhz", "mclk_mhz", "fclk_mhz", "Current Socket Graphics Package Power (W)", "Temperature (Sensor junction) (C)", "Temperature (Sensor memory) (C)", "gpu_still_busy_when_read")}}))
PY
