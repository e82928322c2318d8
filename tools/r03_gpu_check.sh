#!/bin/bash
# Round-3 first GPU pass: parity tests of the shipped library, the default bench line, the old-vs-new quantise kernels.
set -o pipefail
O=gpurun_out/r03a; mkdir -p $O
cd "$(dirname "$0")/.."
echo "== pytest -m gpu" | tee $O/progress.txt
timeout -k 10 900 python -m pytest ${PYTEST_TARGET:-tests} -m gpu -x -q --timeout=240 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/progress.txt
tail -5 $O/pytest_gpu.txt
echo "== bench default" | tee -a $O/progress.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/progress.txt
AB=efficient-llm-inference_amd/lib/ab/libkvq_hip.so
for w in llama3_8b_mixed_seq16k gpt2m_int4_seq4k; do
  for t in "quant_tile=1" "quant_tile=0" "quant_tile=0 quant_geo128=1" "quant_tile=1 quant_tile_tt=4"; do
    args=""; for kv in $t; do args="$args --tunable $kv"; done
    echo "== $w $t" | tee -a $O/progress.txt
    KVQ_HIP_LIB=$AB timeout -k 10 300 python bench.py --steps 10 --warmup 3 --workload $w --rotate-caches 4 --no-cpu-baseline --no-subrecords $args \
      > "$O/bench_${w}_$(echo $t | tr ' =' '__').json" 2>> $O/bench_ab.err || echo "rc=$?" | tee -a $O/progress.txt
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03a/bench_*.json')):
    try: j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'unparsed', e); continue
    rq=j.get('roofline_quantise',{})
    print(f.split('/')[-1], 'value', j.get('value'), 'deq', j['roofline']['avg_launch_ms'], j['roofline']['frac'],
          {k:(v['avg_launch_ms'],v['frac'],v['kernel'][:40]) for k,v in rq.items()})
PY
