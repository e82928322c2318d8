#!/bin/bash
cd /root/repo 2>/dev/null || cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03nb3; mkdir -p $O
AB=efficient-llm-inference_amd/lib/ab/libkvq_hip.so
for rep in 1 2; do for nb in 2 3; do
  line=$(KVQ_HIP_LIB=$AB timeout -k 10 300 python bench.py --steps 30 --warmup 5 --workload llama3_8b_decode_attn_seq16k_b8 --tunable attn_lds_nb=$nb 2>>$O/err.txt | tail -1)
  echo "nb=$nb :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(round(r["avg_launch_ms"]*1e3,2), r["frac"], r["kernel"][:60])')" | tee -a $O/sweep.txt
done; done
