#!/bin/bash
# LDS-staged vs register-staged streaming attention: parity tests, then layer-call time at batch 8 / batch 1 over ring
# depth and wave-slot count (A-B library for the nb / slots keys).
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03b}; mkdir -p $O
echo "== pytest" | tee $O/progress.txt
timeout -k 10 900 python -m pytest tests/test_gpu_attn.py tests/test_gpu_benchmarker.py tests/test_gpu_fullsize.py tests/test_bench_cli.py -m gpu -x -q --timeout=300 > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/progress.txt; tail -4 $O/pytest.txt | tee -a $O/progress.txt
AB=efficient-llm-inference_amd/lib/ab/libkvq_hip.so
run() {  # workload, tunables...
  local w=$1; shift
  local args=""; for kv in "$@"; do args="$args --tunable $kv"; done
  local line
  line=$(KVQ_HIP_LIB=$AB timeout -k 10 300 python bench.py --steps 30 --warmup 5 --workload $w $args 2>>$O/sweep.err | tail -1)
  echo "$w $* :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(r["avg_launch_ms"]*1e3, "us/layer", r["frac"], r["kernel"][:60])' 2>/dev/null || echo FAILED)" | tee -a $O/sweep.txt
}
for w in llama3_8b_decode_attn_seq16k_b8 llama3_8b_decode_attn_seq16k; do
  run $w attn_lds=0
  run $w attn_lds=1
  run $w attn_lds=1 attn_lds_nb=2
  run $w attn_lds=1 attn_lds_nb=4
  run $w attn_lds=1 attn_lds_nb=2 attn_stream_slots=1536
  run $w attn_lds=1 attn_lds_nb=3 attn_stream_slots=768
  run $w attn_lds=1 attn_lds_nb=3 attn_stream_slots=2048
done
run llama2_7b_decode_attn_seq4k_b8 attn_lds=0
