#!/bin/bash
# Loads-only calibration of the streaming decode-attention kernel (batch 8): the shipped library, then the
# KVQ_ATTN_CALIB builds copied over it IN THE BOX'S SCRATCH COPY of the tree (make calib_attn first, here).
L=efficient-llm-inference_amd/lib
wl=llama3_8b_decode_attn_seq16k_b8
run() { timeout -k 10 200 python3 bench.py --workload $wl --steps 20 --warmup 3 "${@:2}" > gpurun_out/r02ab_$1.json 2>/dev/null; python3 -c "import json;j=json.load(open('gpurun_out/r02ab_$1.json'));print('$1', j['ms_per_step'], j['value'], j.get('roofline',{}).get('avg_launch_ms'))"; }
for rep in a b; do
  for roll in 1 0; do
    run real_roll${roll}_$rep --tunable attn_stream_roll=$roll
  done
done
cp $L/libkvq_hip.so /tmp/keep.so
cp $L/calib_attn/libkvq_hip.so $L/libkvq_hip.so
for roll in 1 0; do
  run calib_roll${roll} --tunable attn_stream_roll=$roll
done
cp /tmp/keep.so $L/libkvq_hip.so
