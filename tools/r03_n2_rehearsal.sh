#!/bin/bash
# The N > 1 default line on the 1-GPU box (two ranks share the card, gloo carries the reductions: a rehearsal of the
# code path — sharded_quant sub-record, barriers, max-over-ranks — never a scaling number), and smoke().
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03l}; mkdir -p $O
echo "== smoke" | tee $O/progress.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/progress.txt; tail -2 $O/smoke.log
echo "== bench --gpus 2 rehearsal" | tee -a $O/progress.txt
timeout -k 10 600 python bench.py --gpus 2 --share-gpu --allow-gloo-timing --steps 10 --warmup 3 > $O/bench_n2.json 2> $O/bench_n2.err; echo "n2 rc=$?" | tee -a $O/progress.txt
tail -3 $O/bench_n2.err
python - "$O" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]+'/bench_n2.json').read().strip().splitlines()[-1])
print('n_gpus', j['n_gpus'], 'value', j['value'], 'backend', j['config']['timing_reduction_backend'])
print('sharded_quant', {k: j['sharded_quant'][k] for k in ('value','ms_per_step','kernels')}, j['sharded_quant']['config']['collective_backend'], j['sharded_quant']['config']['layer_chunks'])
print('decode', j['decode'].get('tokens_per_sec'), 'evict', j['configs']['llama3_8b_evict_seq32k'].get('value'), 'cpu_baseline' in j)
PY
