#!/bin/bash
# Round 4, VERDICT item 1: the quantise tile against its row strides (time + DRAM read-credit stalls per stride), the
# outlier-channel input, and the merged-output-piece traffic pattern (kvq_microbench quantwg).
#   -> gpurun_out/r04stride/{sweep_normal.jsonl,sweep_outlier.jsonl,pmc_<pads>.csv,quantwg*.txt,quantpat.txt}
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD; O=$R/gpurun_out/r04stride; mkdir -p $O; export TMPDIR=/tmp
MB=$R/efficient-llm-inference_amd/lib/kvq_microbench
echo "== microbench quantwg (pad 0)" | tee $O/progress.txt
timeout -k 10 300 $MB quantwg 20 > $O/quantwg.txt 2>&1 || exit 1
for P in 16 17 1040; do
  echo "== microbench quantwg pad $P" | tee -a $O/progress.txt
  KVQ_PAD_IN=$P KVQ_PAD_OUT=$P timeout -k 10 300 $MB quantwg 20 > $O/quantwg_pad$P.txt 2>&1 || exit 1
done
echo "== microbench quantpat" | tee -a $O/progress.txt
timeout -k 10 300 $MB quantpat 20 > $O/quantpat.txt 2>&1 || exit 1
echo "== sweep normal" | tee -a $O/progress.txt
timeout -k 10 600 python3 tools/quant_stride_sweep.py > $O/sweep_normal.jsonl 2> $O/sweep_normal.err || exit 1
echo "== sweep outlier" | tee -a $O/progress.txt
timeout -k 10 300 python3 tools/quant_stride_sweep.py --dist outlier --pads 16 > $O/sweep_outlier.jsonl 2> $O/sweep_outlier.err || exit 1
failed=0
for PADS in 0,0 16,0 0,16 16,16 17,17 1040,1040; do
  tag=${PADS/,/_}
  i=0
  for P in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
    i=$((i+1))
    echo "== pmc $PADS pass $i" | tee -a $O/progress.txt
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_${tag}_p$i -- python3 $R/tools/quant_stride_sweep.py --one $PADS --iters 8 > $O/pmc_${tag}_p$i.out 2> $O/pmc_${tag}_p$i.err) || { failed=$((failed+1)); echo "pass FAILED" | tee -a $O/progress.txt; }
  done
done
python3 - "$O" <<'PY'
import csv, glob, os, collections, sys
O = sys.argv[1]
rows = []
for d in sorted(glob.glob(os.path.join(O, "pmc_*_p*"))):
    if not os.path.isdir(d):
        continue
    tag = os.path.basename(d)[4:].rsplit("_p", 1)[0]
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "quant_tile_k" not in r["Kernel_Name"]:
                continue
            kind = "int" + r["Kernel_Name"].split("quant_tile_k<")[1].split(",")[1].strip()  # template argument 2 = BITS
            a = agg[(kind, r["Counter_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for (kind, c), (n, v, t) in sorted(agg.items()):
        rows.append((tag, kind, c, n, v / n, t / n))
with open(os.path.join(O, "pmc_by_stride.csv"), "w") as out:
    out.write("pad_in_pad_out_tokens,kind,counter,launches,mean_per_launch,mean_kernel_ns_in_that_pass\n")
    for r in rows:
        out.write(f"{r[0]},{r[1]},{r[2]},{r[3]},{r[4]:.1f},{r[5]:.0f}\n")
print(open(os.path.join(O, "pmc_by_stride.csv")).read())
PY
find $O -name "*.csv" -path "*pmc_*_p*" -size +2M -delete
echo "failed passes: $failed" | tee -a $O/progress.txt
exit $failed
