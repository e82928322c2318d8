#!/bin/bash
# smoke() + the default bench line as two ranks sharing the one GPU (gloo): a rehearsal of the N > 1 code path after bench.py changes
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04w; mkdir -p $O
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 2 --share-gpu --allow-gloo-timing --steps 10 --warmup 3 > $O/n2.json 2> $O/n2.err || { tail -20 $O/n2.err; exit 1; }
python3 - $O/n2.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["n_gpus"], j["value"], j["sharded_quant"]["config"]["collective_backend"], j["sharded_quant"]["value"], list(j["configs"].keys()),
      j["device_state"]["gpu_still_busy_when_read"], j["run_s"])
PY
