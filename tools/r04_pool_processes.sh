#!/bin/bash
# The pool kernel in four fresh PROCESSES on one box: is the box-to-box spread of chunk_summarize_kv (0.73-0.84) a property of the
# box, or of where a process's 32 GiB land in HBM?   -> gpurun_out/r04v/same_box_four_processes.txt
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04v; mkdir -p $O
for i in 1 2 3 4; do
  timeout -k 10 200 python3 bench.py --workload llama3_8b_evict_seq32k --steps 6 --warmup 2 2>>$O/err.txt | tail -1 | python3 -c "
import json, sys
j = json.loads(sys.stdin.read()); r = j['roofline']; d = j.get('device_state') or {}
print('process $i pool', r['avg_launch_ms'], 'ms', r['frac'], 'min', r['min_launch_ms'], 'max', r['max_launch_ms'], '| window', j['roofline_window']['frac'], '|', d.get('Current Socket Graphics Package Power (W)'), 'W', d.get('Temperature (Sensor memory) (C)'), 'C mem')" | tee -a $O/same_box_four_processes.txt || exit 1
done
