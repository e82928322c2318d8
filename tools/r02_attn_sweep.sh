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
# streaming kernel (one wave per several tiles, next tile in flight): by size, and forced shapes, vs the one-tile kernel
for tc in 64 32; do for tpw in 0 2 4 8; do
  run --workload llama3_8b_decode_attn_seq16k_b8 --steps 20 --warmup 3 --tunable attn_stream_tc=$tc --tunable attn_stream_tpw=$tpw
done; done
run --workload llama3_8b_decode_attn_seq16k_b8 --steps 20 --warmup 3 --tunable attn_stream_tpw=-1
run --workload llama3_8b_decode_attn_seq16k --steps 20 --warmup 3 --tunable attn_stream_tpw=2
run --workload llama3_8b_decode_attn_seq16k --steps 20 --warmup 3 --tunable attn_stream_tpw=2 --tunable attn_stream_tc=32
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
