#!/bin/bash
# SQ / TA / TCC counters of the batch-8 decode-attention workload (LDS-staged kernel): where its waves wait.
set -o pipefail
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/${1:-r03i}; mkdir -p $O; export TMPDIR=/tmp
i=0; failed=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); echo "pass $i: $P" | tee -a $O/progress.txt
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 4 --warmup 2 --workload llama3_8b_decode_attn_seq16k_b8 > $O/p$i.out 2> $O/p$i.err) || { failed=$((failed+1)); echo "pass $i FAILED" | tee -a $O/progress.txt; grep -m2 -i "error\|exceeds" $O/p$i.err | tee -a $O/progress.txt; }
done
python3 - "$O" <<'PY'
import csv, glob, os, collections, sys
O = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for f in glob.glob(os.path.join(O, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "decode_attn" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kvq::", "")
        a = agg[(k, r["Counter_Name"])]
        a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
with open(os.path.join(O, "summary.csv"), "w") as out:
    out.write("kernel,counter,launches,mean_per_launch,mean_kernel_ns_in_that_pass\n")
    for (k, c), (n, v, d) in sorted(agg.items()):
        out.write(f'"{k}",{c},{n},{v / n:.1f},{d / n:.0f}\n')
print(open(os.path.join(O, "summary.csv")).read())
PY
rm -rf $O/p[0-9]*/
echo "failed passes: $failed" | tee -a $O/progress.txt
exit $failed
