#!/bin/bash
# Round 4: the one-pass short-context attention kernel (A-B key attn_onepass) — its tests, then A-B against the two-launch path
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04x; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_attn.py -m ab -x -q -k "one_pass" > $O/pytest_onepass.txt 2>&1 || { tail -30 $O/pytest_onepass.txt; exit 1; }
tail -2 $O/pytest_onepass.txt
for W in gpt2_decode_attn_seq1k llama3_8b_decode_attn_seq1k llama3_8b_decode_attn_seq2k_b8; do
  bash tools/sweep.sh r04x_$W --workload $W --reps 2 --steps 50 --warmup 5 -- "" "attn_onepass=1" || exit 1
done
