#!/bin/bash
# rocprofv3 PMC passes of one bench.py command: HBM traffic (FETCH_SIZE / WRITE_SIZE), L2 (TCC) request shapes and
# stalls, texture-addresser / L1 (TA / TCP) stalls, wave-level (SQ) issue counters. One counter group per pass,
# --kernel-trace only (gpurun refuses --pmc together with the runtime / hip / hsa trace domains). Round 2's version
# put five counters into one TA/TCP pass, the profiler aborted ("error code 38: request exceeds the capabilities of the
# hardware to collect") and the script went on: here TA, TCP and GRBM counters are separate passes and ANY failed pass
# makes the script exit non-zero (after the other passes have run, so one bad group does not cost the rest).
#   usage: tools/r03_pmc.sh <tag> <bench.py args ...>      -> gpurun_out/<tag>/summary.csv
set -o pipefail
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
PASSES=(
  "FETCH_SIZE"
  "WRITE_SIZE"
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_CYCLE_sum"
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
  "TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"
  "GRBM_GUI_ACTIVE"
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES"
)
[ -n "$PMC_ONLY_TRAFFIC" ] && PASSES=("FETCH_SIZE" "WRITE_SIZE")
failed=0; i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  echo "pass $i: $P" | tee -a $O/progress.txt
  (cd /tmp && timeout -k 10 420 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -- python3 $R/bench.py "$@" > $O/p$i.out 2> $O/p$i.err)
  rc=$?
  if [ $rc -ne 0 ]; then failed=$((failed+1)); echo "pass $i FAILED (rc=$rc): $P" | tee -a $O/progress.txt; grep -m3 -i "error\|abort\|exceeds" $O/p$i.err | tee -a $O/progress.txt; fi
done
python3 - "$O" <<'PY'
import csv, glob, os, collections, sys
O = sys.argv[1]
dur = collections.defaultdict(lambda: [0, 0.0])
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(O, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "kvq::" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kvq::", "")
        agg[(k, r["Counter_Name"])][0] += 1
        agg[(k, r["Counter_Name"])][1] += float(r["Counter_Value"])
        dur[(k, r["Counter_Name"])][0] += 1
        dur[(k, r["Counter_Name"])][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
with open(os.path.join(O, "summary.csv"), "w") as out:
    out.write("kernel,counter,launches,mean_per_launch,mean_kernel_ns_in_that_pass\n")
    for (k, c), (n, v) in sorted(agg.items()):
        out.write(f'"{k}",{c},{n},{v / n:.1f},{dur[(k, c)][1] / n:.0f}\n')
# HBM bytes per launch as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE are in KiB, and gfx950 tallies the
# 128-byte requests of wide streaming reads at 64 B: (2 * FETCH_SIZE + WRITE_SIZE) * 1024
kern = sorted({k for (k, c) in agg})
with open(os.path.join(O, "traffic.csv"), "w") as out:
    out.write("kernel,launches,fetch_kib_raw,write_kib,hbm_bytes_per_launch\n")
    for k in kern:
        if (k, "FETCH_SIZE") in agg and (k, "WRITE_SIZE") in agg:
            f = agg[(k, "FETCH_SIZE")][1] / agg[(k, "FETCH_SIZE")][0]
            w = agg[(k, "WRITE_SIZE")][1] / agg[(k, "WRITE_SIZE")][0]
            out.write(f'"{k}",{agg[(k, "FETCH_SIZE")][0]},{f:.1f},{w:.1f},{(2 * f + w) * 1024:.0f}\n')
print(open(os.path.join(O, "traffic.csv")).read())
PY
echo "failed passes: $failed" | tee -a $O/progress.txt
exit $failed
