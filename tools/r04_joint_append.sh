#!/bin/bash
# Round 4: the joint K + V decode append of a batch-sharded job (sharding.quantize_kv_batch_sharded) — tests, the N = 1
# bench line (two_phase sub-record = the phases a rank of a larger job runs) and a 2-rank rehearsal on the one GPU (gloo).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04m; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_sharded_batch.py -m gpu -x -q > $O/pytest_sharded.txt 2>&1 || { tail -30 $O/pytest_sharded.txt; exit 1; }
tail -2 $O/pytest_sharded.txt
for w in llama3_8b_batch64_sharded_append llama3_8b_batch64_sharded_prefill512; do
  timeout -k 10 300 python3 bench.py --workload $w --steps 200 --warmup 20 2>>$O/err.txt | tail -1 | tee -a $O/n1.jsonl || exit 1
done
for w in llama3_8b_batch64_sharded_append llama3_8b_batch64_sharded_prefill512; do
  timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --share-gpu --allow-gloo-timing --workload $w --steps 100 --warmup 10 2>>$O/err.txt | tail -1 | tee -a $O/n2_rehearsal.jsonl || exit 1
done
# A-B: the round-3 plan (per set, CHUNKS_PER_SET layer chunks each on several ranks) on the same rehearsal
KVQ_SHARD_SMALL_TABLE_BYTES=0 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 \
  bench.py --gpus 2 --share-gpu --allow-gloo-timing --workload llama3_8b_batch64_sharded_append --steps 100 --warmup 10 2>>$O/err.txt | tail -1 | tee -a $O/n2_rehearsal_per_set.jsonl || exit 1
KVQ_SHARD_SMALL_TABLE_BYTES=0 timeout -k 10 300 python3 bench.py --workload llama3_8b_batch64_sharded_append --steps 200 --warmup 20 2>>$O/err.txt | tail -1 | tee -a $O/n1_per_set.jsonl
