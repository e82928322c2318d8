#!/bin/bash
# TCC / TA / TCP counters of the default bench's kernels (fused quantise vs dequantise): where the quantise kernel's
# ~10 points of roofline go. Separate PMC passes, --kernel-trace only.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out/r02z; mkdir -p $O; export TMPDIR=/tmp
i=0
for P in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_CYCLE_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
         "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_READ_sum TCC_WRITE_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/p$i.err) || { echo "pass $i failed: $P"; tail -3 $O/p$i.err; }
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.getcwd(), "gpurun_out", "r02z")
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
print(open(os.path.join(O, "summary.csv")).read())
PY
