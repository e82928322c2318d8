#!/bin/bash
# wide quantise tile: 1024-thread workgroups (one per CU) against 512-thread ones (two per CU, half the tile), A-B library
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03x}; mkdir -p $O
AB=efficient-llm-inference_amd/lib/ab/libkvq_hip.so
KVQ_HIP_LIB=$AB timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m ab -x -q --timeout=300 -k "wide" > $O/pytest_ab.txt 2>&1; echo "ab rc=$?" | tee $O/progress.txt; tail -3 $O/pytest_ab.txt | tee -a $O/progress.txt
for rep in 1 2; do for blk in 1024 512; do
  KVQ_HIP_LIB=$AB timeout -k 10 300 python bench.py --steps 10 --warmup 3 --workload llama3_8b_batch64_sharded_prefill512 --tunable quant_wide_blk=$blk > $O/b_$blk_$rep.json 2>> $O/bench.err
  python -c "import json,sys; j=json.loads(open('$O/b_$blk_$rep.json').read().strip().splitlines()[-1]); print('blk $blk', j['value'], j['ms_per_step'], j['kernels'])" | tee -a $O/progress.txt
done; done
