#!/bin/bash
# Round-2 evidence run on the GPU box (through gpurun, from the repo root):
#   rocprofv3 --kernel-trace --stats of the default bench and of the decode-attention workloads,
#   PMC passes (FETCH_SIZE and WRITE_SIZE separately, never combined with trace domains other than
#   --kernel-trace), the N = 2 rehearsal of bench.py's self-launcher, smoke().
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${TAG:-r02p}
export TAG
O=$PWD/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/steps.log; }

log "smoke"; timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo "smoke failed" | tee -a $O/steps.log; exit 1; }

log "default bench (as the driver runs it)"
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || exit 1

log "rocprofv3 kernel stats: headline"
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_headline -- python3 $OLDPWD/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats_headline.err) || exit 1

for wl in llama3_8b_decode_attn_seq16k llama3_8b_decode_attn_seq16k_b8; do
  log "rocprofv3 kernel stats: $wl"
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python3 $OLDPWD/bench.py --workload $wl --steps 10 --warmup 2 > $O/bench_${wl}_under_rocprof.json 2> $O/stats_$wl.err) || exit 1
  timeout -k 10 300 python3 bench.py --workload $wl --steps 20 --warmup 3 > $O/bench_$wl.json 2>> $O/bench_attn.err || exit 1
done

log "evict bench + stats"
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_evict -- python3 $OLDPWD/bench.py --workload llama3_8b_evict_seq32k --steps 5 --warmup 2 > $O/bench_evict_under_rocprof.json 2> $O/stats_evict.err) || exit 1

for c in FETCH_SIZE WRITE_SIZE; do
  log "PMC $c: headline"
  (cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_headline -- python3 $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/pmc_${c}_headline.json 2> $O/pmc_${c}_headline.err) || exit 1
  log "PMC $c: attention b8"
  (cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_attn_b8 -- python3 $OLDPWD/bench.py --workload llama3_8b_decode_attn_seq16k_b8 --steps 3 --warmup 1 > $O/pmc_${c}_attn_b8.json 2> $O/pmc_${c}_attn_b8.err) || exit 1
done

log "N = 2 rehearsal of the self-launcher (two ranks share the one GPU)"
timeout -k 10 400 python3 bench.py --gpus 2 --steps 10 --warmup 3 --share-gpu --allow-gloo-timing --no-cpu-baseline > $O/n2_rehearsal.json 2> $O/n2_rehearsal.err; echo "rc=$?" >> $O/n2_rehearsal.err
timeout -k 10 200 python3 bench.py --gpus 2 --steps 3 --warmup 1 > $O/n2_no_flags.out 2> $O/n2_no_flags.err; echo "rc=$?" >> $O/n2_no_flags.err
timeout -k 10 200 python3 bench.py --gpus 2 --steps 3 --warmup 1 --share-gpu > $O/n2_share_only.out 2> $O/n2_share_only.err; echo "rc=$?" >> $O/n2_share_only.err

log "summaries"
python3 - <<'PY'
import csv, glob, json, os, collections
O = os.path.join(os.getcwd(), "gpurun_out", os.environ.get("TAG", "r02p"))
def stats(d):
    f = glob.glob(os.path.join(O, d, "**", "*kernel_stats.csv"), recursive=True)
    return f[0] if f else None
for d in sorted(os.listdir(O)):
    if d.startswith("stats_") and os.path.isdir(os.path.join(O, d)):
        f = stats(d)
        if f:
            rows = list(csv.reader(open(f)))
            with open(os.path.join(O, d + "_kernel_stats.csv"), "w") as out:
                w = csv.writer(out, quoting=csv.QUOTE_ALL)
                for r in rows[:1] + [r for r in rows[1:] if "kvq::" in r[0] or "attn_fwd" in r[0]]:
                    w.writerow(r)
def pmc(d):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "kvq::" not in r["Kernel_Name"]:
                continue
            k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg
with open(os.path.join(O, "pmc_summary.csv"), "w") as out:
    out.write("run,kernel,counter,launches,mean_KiB_per_launch\n")
    for d in sorted(os.listdir(O)):
        if d.startswith("pmc_") and os.path.isdir(os.path.join(O, d)):
            for (k, c), (n, v) in sorted(pmc(d).items()):
                out.write(f'{d},"{k}",{c},{n},{v / n:.1f}\n')
print(open(os.path.join(O, "pmc_summary.csv")).read())
for f in sorted(glob.glob(os.path.join(O, "stats_*_kernel_stats.csv"))):
    print("##", os.path.basename(f)); print(open(f).read())
PY
for f in bench_default.json bench_llama3_8b_decode_attn_seq16k.json bench_llama3_8b_decode_attn_seq16k_b8.json n2_rehearsal.json; do echo "## $f"; cut -c1-1500 $O/$f; done
tail -3 $O/n2_no_flags.err $O/n2_share_only.err $O/n2_rehearsal.err
