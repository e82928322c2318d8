set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04g; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || { tail -40 $O/pytest_gpu.txt; exit 1; }
tail -3 $O/pytest_gpu.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_attn.py -x -q -m ab > $O/pytest_ab_attn.txt 2>&1 || { tail -40 $O/pytest_ab_attn.txt; exit 1; }
tail -3 $O/pytest_ab_attn.txt
for W in llama3_8b_decode_attn_seq16k llama3_8b_decode_attn_seq16k_b8 gpt2_decode_attn_seq1k llama2_7b_decode_attn_seq4k_b8 llama32_1b_decode_attn_seq16k_b8; do
  timeout -k 10 300 python3 bench.py --workload $W --steps 30 --warmup 5 > $O/$W.json 2> $O/$W.err || exit 1
  python3 -c "
import json; j=json.loads(open('$O/$W.json').read().strip().splitlines()[-1]); r=j['roofline']; print('$W', round(r['avg_launch_ms']*1e3,2), r['frac'], r['kernel'][:90])"
done
timeout -k 10 600 python3 bench.py --workload decode:gpt2:quant_int8:512:512 --steps 1 --warmup 1 > $O/decode.json 2> $O/decode.err || exit 1
python3 -c "
import json; j=json.loads(open('$O/decode.json').read().strip().splitlines()[-1]); print({k:j.get(k) for k in ('value','fused_attention_tokens_per_sec','graph_decode_tokens_per_sec','full_cache_tokens_per_sec','est_kv_cache_mb')})"
