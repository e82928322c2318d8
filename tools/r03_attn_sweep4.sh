#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03j}; mkdir -p $O
echo "== pytest ab (lds)" | tee $O/progress.txt
timeout -k 10 600 python -m pytest tests/test_gpu_attn.py -m ab -x -q --timeout=300 -k "lds_staged_kernels_equal" > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/progress.txt; tail -3 $O/pytest.txt | tee -a $O/progress.txt
AB=efficient-llm-inference_amd/lib/ab/libkvq_hip.so
run() {
  local w=$1; shift
  local args=""; for kv in "$@"; do args="$args --tunable $kv"; done
  local line
  line=$(KVQ_HIP_LIB=$AB timeout -k 10 300 python bench.py --steps 30 --warmup 5 --workload $w $args 2>>$O/sweep.err | tail -1)
  echo "$w $* :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(round(r["avg_launch_ms"]*1e3,2), "us/layer", r["frac"], r["kernel"][:60])' 2>/dev/null || echo FAILED)" | tee -a $O/sweep.txt
}
w=llama3_8b_decode_attn_seq16k_b8
run $w
run $w attn_lds_nb=1 attn_stream_slots=2048
run $w attn_lds_nb=1 attn_stream_slots=3072
run $w attn_lds_nb=1 attn_stream_slots=1536
run $w attn_lds_nb=1 attn_stream_slots=1024
run $w attn_lds_nb=1 attn_stream_slots=2560
w=llama3_8b_decode_attn_seq16k
run $w
run $w attn_lds=1 attn_stream_tpw=1 attn_lds_nb=1
run $w attn_lds=1 attn_stream_tpw=2 attn_lds_nb=1
