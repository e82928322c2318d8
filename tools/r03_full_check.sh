#!/bin/bash
# Whole-tree GPU check: the shipped library's tests, the A-B library's variant-equality tests, the default bench line,
# the decode-attention workloads.
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03e}; mkdir -p $O
echo "== pytest -m gpu" | tee $O/progress.txt
( time timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout=300 ) > $O/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?" | tee -a $O/progress.txt; tail -6 $O/pytest_gpu.txt | tee -a $O/progress.txt
if [ -z "$SKIP_AB" ]; then
echo "== pytest -m ab" | tee -a $O/progress.txt
( time timeout -k 10 1100 python -m pytest tests -m ab -x -q --timeout=300 ) > $O/pytest_ab.txt 2>&1; echo "pytest ab rc=$?" | tee -a $O/progress.txt; tail -6 $O/pytest_ab.txt | tee -a $O/progress.txt
fi
echo "== bench default" | tee -a $O/progress.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/progress.txt
for w in llama3_8b_decode_attn_seq16k_b8 llama3_8b_decode_attn_seq16k llama2_7b_decode_attn_seq4k_b8 llama32_1b_decode_attn_seq16k_b8 gpt2_decode_attn_seq1k; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --workload $w > $O/bench_$w.json 2>> $O/bench_attn.err; echo "$w rc=$?" | tee -a $O/progress.txt
done
python - "$O" <<'PY'
import json,glob,sys
O=sys.argv[1]
for f in sorted(glob.glob(O+'/bench_*.json')):
    try: j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'unparsed', e); continue
    r=j['roofline']; print(f.split('/')[-1], 'value', j.get('value'), r['avg_launch_ms'], r['frac'], r['kernel'][:90])
PY
