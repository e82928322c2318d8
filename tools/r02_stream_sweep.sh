#!/bin/bash
# Round-2: the streaming decode-attention kernel vs the one-tile kernel at batch 8 / batch 1 (GPU box).
set -o pipefail
out=gpurun_out/r02_stream_sweep.jsonl
: > $out
run() { echo "## $*" >> $out; timeout -k 10 300 python3 bench.py "$@" >> $out 2>> gpurun_out/r02_stream_sweep.err || echo "FAILED rc=$? $*" >> $out; }
run --workload llama3_8b_decode_attn_seq16k_b8 --steps 20 --warmup 3 --tunable attn_stream_tpw=-1
for tc in 64 32; do for tpw in 0 2 3 4 6 8; do
  run --workload llama3_8b_decode_attn_seq16k_b8 --steps 20 --warmup 3 --tunable attn_stream_tc=$tc --tunable attn_stream_tpw=$tpw
done; done
run --workload llama3_8b_decode_attn_seq16k_b8 --steps 20 --warmup 3 --tunable attn_stream_tpw=-1
run --workload llama3_8b_decode_attn_seq16k --steps 20 --warmup 3
run --workload llama3_8b_decode_attn_seq16k --steps 20 --warmup 3 --tunable attn_stream_tpw=2
run --workload llama3_8b_decode_attn_seq16k --steps 20 --warmup 3 --tunable attn_stream_tpw=2 --tunable attn_stream_tc=32
run --workload llama3_8b_decode_attn_seq16k --steps 20 --warmup 3 --tunable attn_stream_tpw=1 --tunable attn_stream_tc=32
python3 - <<'PY'
import json
for ln in open("gpurun_out/r02_stream_sweep.jsonl"):
    if ln.startswith("##") or ln.startswith("FAILED"):
        print(ln.strip()); continue
    try: d = json.loads(ln)
    except Exception: continue
    r = d["roofline"]
    print(f'   -> {d["ms_per_step"]*1e3/d["config"]["shape_L_B_Hq_Hkv_T_D"][0]:.2f} us/layer  {r["achieved"]} GB/s  frac {r["frac"]}')
PY
