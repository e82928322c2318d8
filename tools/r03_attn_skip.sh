#!/bin/bash
# where the ring kernel's tile time goes (make calib_attn_skip; inexact results): the shipped kernel against builds without the score
# MFMAs (1), without the P·V MFMAs (2), without both (3), without the V conversion (4), without exp2 (8), without all four (15)
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03skip}; mkdir -p $O
L=efficient-llm-inference_amd/lib
for rep in 1 2; do for v in shipped skip1 skip2 skip3 skip4 skip8 skip15; do
  lib=$L/$v/libkvq_hip.so; [ $v = shipped ] && lib=$L/libkvq_hip.so
  line=$(KVQ_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 30 --warmup 5 --workload llama3_8b_decode_attn_seq16k_b8 2>>$O/err.txt | tail -1)
  echo "$v :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(round(r["avg_launch_ms"]*1e3,2), "us per layer call")')" | tee -a $O/sweep.txt
done; done
