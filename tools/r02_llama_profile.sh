#!/bin/bash
# Llama-3-8B architecture, 16K prompt: per-token time of each cache path + rocprofv3 kernel table of the graph path.
# Run on the GPU box from the repo root.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 600 python3 tools/llama_decode_phases.py phases full fullctx staged fused graph > gpurun_out/r02w_llama8b_phases.txt 2>gpurun_out/r02w_llama8b_phases.err
cat gpurun_out/r02w_llama8b_phases.txt
for p in graph; do
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$p -- python3 $R/tools/llama_decode_phases.py $p --new 32 > /dev/null 2>&1)
  f=$(find /tmp/prof_$p -name '*kernel_stats.csv' | head -1)
  head -30 "$f" > gpurun_out/r02w_llama8b_${p}_kernel_stats_top.csv
done
