#!/bin/bash
# the LDS-staged attention kernel with its arithmetic removed (make calib_attn: every loaded word is consumed, nothing is
# computed; results are wrong by construction): the time of its data path alone
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03n}; mkdir -p $O
for lib in efficient-llm-inference_amd/lib/libkvq_hip.so efficient-llm-inference_amd/lib/calib_attn/libkvq_hip.so efficient-llm-inference_amd/lib/libkvq_hip.so efficient-llm-inference_amd/lib/calib_attn/libkvq_hip.so; do
  line=$(KVQ_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 30 --warmup 5 --workload llama3_8b_decode_attn_seq16k_b8 2>>$O/err.txt | tail -1)
  echo "$lib :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(round(r["avg_launch_ms"]*1e3,2), "us/layer", r["frac"], r["kernel"][:60])')" | tee -a $O/calib.txt
done
