#!/bin/bash
# wave priority in the quantise tile kernel (make calib_prio): shipped vs prio1 (high once the loads are issued) vs prio2 (high while issuing them)
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03prio}; mkdir -p $O
L=efficient-llm-inference_amd/lib
for rep in 1 2 3; do for v in shipped prio1 prio2; do
  lib=$L/$v/libkvq_hip.so; [ $v = shipped ] && lib=$L/libkvq_hip.so
  line=$(KVQ_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-subrecords --no-cpu-baseline 2>>$O/err.txt | tail -1)
  echo "$v :: $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); q=j["roofline_quantise"]; print({k:(round(v["avg_launch_ms"]*1e3,1), v["frac"]) for k,v in q.items()})')" | tee -a $O/prio.txt
done; done
