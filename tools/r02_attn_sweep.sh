#!/bin/bash
# Round-2 decode-attention sweep on the GPU box: fused single launch shapes vs the two-launch path.
# Usage (from the repo root, through gpurun): bash tools/r02_attn_sweep.sh
set -o pipefail
out=gpurun_out/r02_attn_sweep.jsonl
: > $out
run() { echo "## $*" >> $out; timeout -k 10 300 python3 bench.py "$@" >> $out 2>> gpurun_out/r02_attn_sweep.err || echo "FAILED rc=$? $*" >> $out; }
for wl in llama3_8b_decode_attn_seq16k llama3_8b_decode_attn_seq16k_b8; do
  run --workload $wl --steps 20 --warmup 3 
  run --workload $wl --steps 20 --warmup 3 --per-layer-calls
  run --workload $wl --steps 20 --warmup 3 --tunable attn_mfma_tc=64
  for shape in "128 4" "128 8" "64 8" "32 16"; do
    set -- $shape
    run --workload $wl --steps 20 --warmup 3 --tunable attn_fused=1 --tunable attn_fused_tc=$1 --tunable attn_fused_nw=$2
  done
  run --workload $wl --steps 20 --warmup 3 --per-layer-calls
  run --workload $wl --steps 20 --warmup 3 --tunable attn_mfma_tc=64
done
run --workload llama32_1b_decode_attn_seq16k_b8 --steps 20 --warmup 3
run --workload llama32_1b_decode_attn_seq16k_b8 --steps 20 --warmup 3 
python3 - <<'PY'
import json
for ln in open("gpurun_out/r02_attn_sweep.jsonl"):
    if ln.startswith("##") or ln.startswith("FAILED"):
        print(ln.strip()); continue
    try: d = json.loads(ln)
    except Exception: continue
    r = d["roofline"]
    print(f'   -> {d["ms_per_step"]*1e3/d["config"]["shape_L_B_Hq_Hkv_T_D"][0]:.2f} us/layer  {r["achieved"]} GB/s  frac {r["frac"]}')
PY
