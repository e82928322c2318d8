#!/bin/bash
# The single-pass wide quantise tile: parity tests, the batch-64 prefill workload (single pass vs the two phases), then the whole
# shipped suite with per-test durations and the default bench line (phases_s).
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03w}; mkdir -p $O
echo "== parity" | tee $O/progress.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_sharded_batch.py -m gpu -x -q --timeout=300 > $O/pytest_parity.txt 2>&1; rc=$?
echo "parity rc=$rc" | tee -a $O/progress.txt; tail -5 $O/pytest_parity.txt | tee -a $O/progress.txt
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --workload llama3_8b_batch64_sharded_prefill512 > $O/bench_sharded_$rep.json 2>> $O/bench.err; echo "sharded $rep rc=$?" | tee -a $O/progress.txt
  python -c "import json,sys; j=json.loads(open('$O/bench_sharded_$rep.json').read().strip().splitlines()[-1]); print('single', j['value'], j['ms_per_step'], j['kernels']); print('two_phase', j['two_phase']['value'], j['two_phase']['ms_per_step'], j['two_phase']['kernels'])" | tee -a $O/progress.txt
done
echo "== pytest -m gpu" | tee -a $O/progress.txt
( time timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout=300 --durations=30 ) > $O/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?" | tee -a $O/progress.txt; tail -45 $O/pytest_gpu.txt | tee -a $O/progress.txt
echo "== bench default" | tee -a $O/progress.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/progress.txt
python -c "import json; j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(j['value'], j['roofline']['frac'], j['run_s'], j['phases_s'])" | tee -a $O/progress.txt
