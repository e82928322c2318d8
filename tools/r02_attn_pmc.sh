#!/bin/bash
# SQ / SQC counters of the batch-8 decode-attention partial kernels (one-tile and streaming), two PMC passes each.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD; O=$R/gpurun_out/r02w; mkdir -p $O; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES"
P2="SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_VALU"
P3="SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
for v in onetile stream; do
  t=$([ $v = onetile ] && echo -1 || echo 0)
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/${v}_p$i -- python3 $R/bench.py --workload llama3_8b_decode_attn_seq16k_b8 --steps 2 --warmup 1 --tunable attn_stream_tpw=$t > /dev/null 2> $O/${v}_p$i.err) || { echo "pass $v $i failed"; tail -3 $O/${v}_p$i.err; }
  done
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.getcwd(), "gpurun_out", "r02w")
with open(os.path.join(O, "summary.csv"), "w") as out:
    out.write("variant,kernel,counter,launches,mean_per_launch\n")
    for d in sorted(os.listdir(O)):
        if not os.path.isdir(os.path.join(O, d)):
            continue
        agg = collections.defaultdict(lambda: [0, 0.0])
        for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "decode_attn" not in r["Kernel_Name"]:
                    continue
                k = (r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kvq::", ""), r["Counter_Name"])
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
        for (k, c), (n, v) in sorted(agg.items()):
            out.write(f'{d},"{k}",{c},{n},{v / n:.1f}\n')
print(open(os.path.join(O, "summary.csv")).read())
PY
