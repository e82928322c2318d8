"""SURVEY §8e's one exchange step: a batched ``[B>1,H,1,D]`` slice whose batch rows are split over
ranks keeps ONE scale per token (reference src/quantization/ops.py:27,48), so the ``[G,T]`` abs-max
table crosses ranks (``all_reduce(MAX)``) between the reduction and the quantisation.

CPU (``-m "not gpu"``): world-size-2 gloo harness — rows sharded by ``shard_batch_rows``, the table
reduced by ``sharding.all_reduce_absmax``, both phases evaluated by the oracle: bit-exact with the
oracle quantising the un-sharded batch, on both ranks.
GPU (``-m gpu``): the same through the HIP kernels (kvq_absmax_tokens / kvq_quant_tokens_from_absmax),
single process (pair == fused kernels) and two processes sharing the card (gloo carries the table).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import kvq_oracle as O
from tests.conftest import ROOT
from tests.util import bits, odt, seeded_kv, to_numpy, to_torch

SHAPE = (3, 5, 4, 9, 64)  # G, B, H, T, D: 5 batch rows over 2 ranks = 3 + 2
APPEND = (6, 5, 8, 1, 128)  # a decode step's K (or V) slices of 6 layers, batch 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(dtype="f16", shape=SHAPE):
    x = seeded_kv(shape, dtype, seed=77, dist="heavy")
    # make the abs-max of some tokens live on rank 1's rows only and of others on rank 0's only
    x32 = O._widen(x, odt(dtype)).copy()
    x32[:, 4, :, 0::2, :] *= 5.0
    x32[:, 0, :, 1::2, :] *= 7.0
    return O.f32_to_bf16_bits(x32) if dtype == "bf16" else x32.astype(x.dtype)


def _cpu_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from efficient_llm_inference_amd import sharding
    sharding.init_distributed(rank, world, None)
    try:
        x = _batch()
        rows = sharding.shard_batch_rows(x.shape[1])
        mine = x[:, rows.start:rows.stop]
        res = {}
        for kind in ("int8", "int4"):
            amax = torch.from_numpy(O.absmax_tokens(mine))
            sharding.all_reduce_absmax(amax)  # gloo: the table is a host tensor here
            qq, _, s32 = O.quantize_tokens_with_absmax(mine, amax.numpy(), kind)
            res[kind] = (qq, s32)
        q.put((rank, list(rows), res))
    finally:
        sharding.shutdown()


def test_sharded_batch_scales_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = _batch()
    assert out[0][1] == [0, 1, 2] and out[1][1] == [3, 4]
    for kind in ("int8", "int4"):
        q_ref, _, s32_ref = O.quantize_tokens(x, kind)
        got = np.concatenate([out[0][2][kind][0], out[1][2][kind][0]], axis=1)
        assert np.array_equal(got, q_ref)
        for r in range(2):  # identical, whole-batch scales on every rank
            assert np.array_equal(out[r][2][kind][1].view(np.uint32), s32_ref.view(np.uint32))
        # and they differ from what each rank would have got on its own rows (the exchange matters)
        assert not np.array_equal(O.quantize_tokens(x[:, :3], kind)[2].view(np.uint32), s32_ref.view(np.uint32))


# ---------------------------------------------------------------------------- GPU


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [SHAPE, (2, 2, 8, 33, 128), (2, 3, 3, 7, 5), (1, 4, 2, 6, 24), (130, 1, 2, 3, 16),
                                   (2, 3, 4, 21, 64), (3, 1, 5, 9, 128), (1, 9, 8, 4, 64)])  # one-wave tile phases: ragged row groups / tokens
@pytest.mark.parametrize("dtype", ["f16", "bf16", "f32"])
def test_split_phases_equal_fused_kernels(shape, dtype):
    """One rank: kvq_absmax_tokens + kvq_quant_tokens_from_absmax == the oracle == kvq_quant_*_tokens, bit for bit
    (vector and generic paths, list and single-buffer inputs, > 128 groups)."""
    from efficient_llm_inference_amd import kernels as K
    G, B, H, T, D = shape
    x_np = seeded_kv(shape, dtype, seed=3, dist="heavy")
    x = to_torch(x_np, dtype)
    amax = K.absmax_tokens(x)
    assert np.array_equal(bits(amax), bits(O.absmax_tokens(x_np, odt(dtype))))
    # kvq_absmax_tokens_acc: no fill launch, the result is the max with what the caller's table holds
    acc = torch.zeros_like(amax)
    K.absmax_tokens(x, acc, accumulate=True)
    assert torch.equal(acc, amax)
    floor = torch.full_like(amax, float(amax.median()))
    K.absmax_tokens(x, floor, accumulate=True)
    assert torch.equal(floor, torch.maximum(amax, torch.full_like(amax, float(amax.median()))))
    for kind in ("int8", "int4"):
        q_ref, _, s32_ref = O.quantize_tokens(x_np, kind, dtype=odt(dtype))
        for src in (x, [x[g] for g in range(G)] if G <= 256 else x):
            q, sc = K.quant_tokens_with_absmax(src, amax, kind)
            assert np.array_equal(to_numpy(q), q_ref) and np.array_equal(bits(sc), bits(s32_ref)), kind
        store = torch.zeros_like(q)
        scales = torch.zeros_like(sc)
        K.quant_tokens(x, store, scales, torch.empty(G * T, dtype=torch.float32, device="cuda"), kind)
        assert torch.equal(store, q) and torch.equal(scales, sc)


def _count_calls(sharding, monkeypatch=None):
    """Count the abs-max launches, table exchanges and quantise launches `sharding` makes from here on."""
    from efficient_llm_inference_amd import kernels as K
    calls = {"absmax": 0, "all_reduce": 0, "quant": 0}

    def counted(fn, key):
        def wrapper(*a, **kw):
            calls[key] += 1
            return fn(*a, **kw)
        return wrapper

    setter = monkeypatch.setattr if monkeypatch is not None else setattr
    setter(K, "absmax_tokens", counted(K.absmax_tokens, "absmax"))
    setter(K, "quant_tokens_with_absmax", counted(K.quant_tokens_with_absmax, "quant"))
    setter(sharding, "all_reduce_absmax", counted(sharding.all_reduce_absmax, "all_reduce"))
    return calls


def _gpu_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from efficient_llm_inference_amd import _lib, sharding
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    # both ranks share the one GPU of the box: RCCL refuses duplicate devices, gloo carries the table
    backend = sharding.init_distributed(rank, world, dev, allow_gloo=True, ranks_share_device=True)
    try:
        res = {}
        for dtype in ("f16", "bf16"):
            x = _batch(dtype, (3, 5, 4, 9, 128))
            rows = sharding.shard_batch_rows(x.shape[1])
            mine = to_torch(np.ascontiguousarray(x[:, rows.start:rows.stop]), dtype)
            for kind in ("int8", "int4"):
                qq, sc = sharding.quantize_tokens_batch_sharded(mine, kind)
                res[(dtype, kind)] = (to_numpy(qq), to_numpy(sc))
        # a decode append: K and V of one token, ONE abs-max launch over both sets and ONE exchange of the joint table
        xk, xv = _batch("f16", APPEND), _batch("f16", APPEND)[::-1].copy()
        rows = sharding.shard_batch_rows(APPEND[1])
        mk, mv = (to_torch(np.ascontiguousarray(a[:, rows.start:rows.stop]), "f16") for a in (xk, xv))
        calls = _count_calls(sharding)
        (qk, sk), (qv, sv) = sharding.quantize_kv_batch_sharded(mk, mv, ("int8", "int4"))
        torch.cuda.synchronize()
        res["append"] = (to_numpy(qk), to_numpy(sk), to_numpy(qv), to_numpy(sv), dict(calls))
        q.put((rank, backend, res))
    finally:
        sharding.shutdown()


@pytest.mark.gpu
def test_sharded_batch_two_processes_share_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=600) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert out[0][1] == out[1][1] == "gloo"
    for dtype in ("f16", "bf16"):
        x = _batch(dtype, (3, 5, 4, 9, 128))
        for kind in ("int8", "int4"):
            q_ref, _, s32_ref = O.quantize_tokens(x, kind, dtype=odt(dtype))
            got = np.concatenate([out[0][2][(dtype, kind)][0], out[1][2][(dtype, kind)][0]], axis=1)
            assert np.array_equal(got, q_ref), (dtype, kind)
            for r in range(2):
                assert np.array_equal(out[r][2][(dtype, kind)][1].view(np.uint32), s32_ref.view(np.uint32)), (dtype, kind, r)
    xk, xv = _batch("f16", APPEND), _batch("f16", APPEND)[::-1].copy()
    for i, (x, kind) in enumerate(((xk, "int8"), (xv, "int4"))):
        q_ref, _, s32_ref = O.quantize_tokens(x, kind, dtype=odt("f16"))
        assert np.array_equal(np.concatenate([out[0][2]["append"][2 * i], out[1][2]["append"][2 * i]], axis=1), q_ref), kind
        for r in range(2):
            assert np.array_equal(out[r][2]["append"][2 * i + 1].view(np.uint32), s32_ref.view(np.uint32)), (kind, r)
    for r in range(2):  # per rank, for the two sets: ONE abs-max launch, ONE exchange, two quantise launches
        assert out[r][2]["append"][4] == {"absmax": 1, "all_reduce": 1, "quant": 2}, out[r][2]["append"][4]


def _rccl_worker(port, q):
    try:
        _rccl_worker_body(port, q)
    except BaseException as exc:  # noqa: BLE001 - the parent must not wait 10 minutes for a result that never comes
        import traceback
        q.put(("error", traceback.format_exc()[-1500:], [], {"exc": repr(exc)}))
        raise


def _rccl_worker_body(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from efficient_llm_inference_amd import sharding
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    backend = sharding.init_distributed(0, 1, dev)  # no allow_gloo: RCCL must really come up
    t = torch.full((4,), 3.0, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=sharding._STATE["group"])
    sharding.barrier()
    # the layer-chunked quantise with its all_reduce(MAX) over RCCL on the side stream (forced: one rank): 5 chunks of
    # one layer each, pipelined abs-max / reduce / quantise, reused buffers, two steps
    ok = {}
    x = _batch("f16", (5, 5, 8, 70, 128))
    xt = to_torch(x, "f16")
    for kind in ("int8", "int4"):
        bufs = sharding.ShardedQuantBuffers(xt, kind, chunk_bytes=1, force_overlap=True)
        for _ in range(2):
            qq, sc = sharding.quantize_tokens_batch_sharded(xt, kind, out=bufs)
        torch.cuda.synchronize()
        q_ref, _, s32_ref = O.quantize_tokens(x, kind, dtype=odt("f16"))
        ok[kind] = bool(bufs.overlap and bufs.n_chunks == 5 and np.array_equal(to_numpy(qq), q_ref)
                        and np.array_equal(to_numpy(sc).view(np.uint32), s32_ref.view(np.uint32)))
    q.put((backend, sharding.backend(), t.cpu().tolist(), ok))
    sharding.shutdown()


@pytest.mark.gpu
def test_rccl_group_comes_up_on_one_rank():
    """The RCCL branch of sharding.init_distributed on the real stack (one rank is all a 1-GPU box offers):
    gloo control group + an RCCL group bound to the device, a reduction through it, barrier, shutdown."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    backend, reported, vals, chunked = q.get(timeout=300)
    assert backend != "error", reported
    p.join(timeout=120)
    assert p.exitcode == 0 and backend == reported == "nccl" and vals == [3.0] * 4
    assert chunked == {"int8": True, "int4": True}, chunked


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["int8", "int4"])
def test_single_rank_takes_one_pass_and_equals_the_two_phases(kind):
    """One rank, nothing to exchange: `quantize_tokens_batch_sharded` takes the un-sharded single-pass call (the 1024-thread
    register tile for a batch-64 slice); `two_phase=True` runs the abs-max and quantise phases a rank of a larger job runs.
    Same bytes, same stored scales, both equal to the oracle; the kernel log tells the two apart."""
    from efficient_llm_inference_amd import _lib, sharding
    shape = (2, 64, 8, 17, 128)
    x = _batch("f16", shape)
    xt = to_torch(x, "f16")
    q_ref, _, s32_ref = O.quantize_tokens(x, kind, dtype=odt("f16"))
    bufs = sharding.ShardedQuantBuffers(xt, kind)
    logs = {}
    for two_phase in (None, True):
        bufs.q.zero_()
        bufs.scales.zero_()
        _lib.kernel_log_clear()
        qq, sc = sharding.quantize_tokens_batch_sharded(xt, kind, out=bufs, two_phase=two_phase)
        torch.cuda.synchronize()
        logs[two_phase] = _lib.kernel_log()
        assert np.array_equal(to_numpy(qq), q_ref) and np.array_equal(to_numpy(sc).view(np.uint32), s32_ref.view(np.uint32)), (two_phase, logs)
    assert len(logs[None]) == 1 and logs[None][0].startswith("quant_wide_k<"), logs
    assert [k.rsplit(", ", 1)[-1] for k in logs[True]] == ["1>", "2>"] and all(k.startswith("quant_tile_k<") for k in logs[True]), logs


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [APPEND, (32, 5, 8, 1, 128), (4, 5, 3, 5, 24), (64, 6, 2, 2, 64)])
@pytest.mark.parametrize("as_list", [False, True])
def test_joint_kv_append_equals_the_per_set_calls(shape, as_list, monkeypatch):
    """`quantize_kv_batch_sharded`: K and V of a decode step (or a few tokens) share ONE abs-max launch and ONE exchange.
    Same bytes and stored scales as the per-set two-phase calls and as the oracle on the un-sharded slice; reused buffers;
    a pair whose strides differ, or whose table is large, falls back to the per-set pipeline."""
    from efficient_llm_inference_amd import _lib, sharding
    G = shape[0]
    xk, xv = _batch("f16", shape), (_batch("f16", shape)[::-1] * np.float16(0.5)).copy()
    tk, tv = to_torch(xk, "f16"), to_torch(xv, "f16")
    if as_list:  # the legacy-tuple form: separately allocated per-layer tensors
        tk, tv = [t.clone() for t in tk], [t.clone() for t in tv]
    outs = (sharding.ShardedQuantBuffers(tk, "int8"), sharding.ShardedQuantBuffers(tv, "int4"))
    assert outs[0].n_chunks == 1 and sharding.kv_joint_table_ok(G, shape[3])
    calls = _count_calls(sharding, monkeypatch)
    for _ in range(2):
        (qk, sk), (qv, sv) = sharding.quantize_kv_batch_sharded(tk, tv, ("int8", "int4"), outs=outs, two_phase=True)
        torch.cuda.synchronize()
    assert calls == {"absmax": 2, "all_reduce": 2, "quant": 4}, calls  # per call: ONE abs-max launch, ONE exchange
    for x, kind, qq, sc in ((xk, "int8", qk, sk), (xv, "int4", qv, sv)):
        q_ref, _, s32_ref = O.quantize_tokens(x, kind, dtype=odt("f16"))
        assert np.array_equal(to_numpy(qq), q_ref), kind
        assert np.array_equal(to_numpy(sc).view(np.uint32), s32_ref.view(np.uint32)), kind
    # fallbacks: one rank with nothing to exchange (two_phase None -> the single-pass kernels), and a strided V
    _lib.kernel_log_clear()
    (qk2, sk2), (qv2, sv2) = sharding.quantize_kv_batch_sharded(tk, tv, ("int8", "int4"), two_phase=None)
    torch.cuda.synchronize()
    assert len(_lib.kernel_log()) == 2 and torch.equal(qk2, qk) and torch.equal(qv2, qv) and torch.equal(sv2, sv) and torch.equal(sk2, sk)
    if not as_list:
        wide = torch.zeros(shape[:3] + (shape[3] + 3, shape[4]), dtype=torch.float16, device="cuda")
        wide[:, :, :, :shape[3]] = tv
        calls.update(absmax=0, all_reduce=0, quant=0)
        (qk3, sk3), (qv3, sv3) = sharding.quantize_kv_batch_sharded(tk, wide[:, :, :, :shape[3]], ("int8", "int4"), two_phase=True)
        torch.cuda.synchronize()
        assert calls["absmax"] == 2 and torch.equal(qv3, qv) and torch.equal(sv3, sv) and torch.equal(qk3, qk), calls
        # V with the same inner strides but every second layer of a larger stack: still ONE abs-max launch (pointer-list form)
        twice = torch.zeros((2 * G,) + tuple(shape[1:]), dtype=torch.float16, device="cuda")
        twice[::2] = tv
        calls.update(absmax=0, all_reduce=0, quant=0)
        (qk4, sk4), (qv4, sv4) = sharding.quantize_kv_batch_sharded(tk, twice[::2], ("int8", "int4"), two_phase=True)
        torch.cuda.synchronize()
        assert calls == {"absmax": 1, "all_reduce": 1, "quant": 2} and torch.equal(qv4, qv) and torch.equal(sv4, sv) and torch.equal(qk4, qk), calls


def test_small_tables_are_exchanged_whole():
    """The chunk plan: a table of at most SMALL_TABLE_BYTES is one chunk (one collective) whatever the rank count;
    the joint K + V path needs the pair's table under the same bound and 2G tensors within one launch's pointer table."""
    from efficient_llm_inference_amd import sharding
    assert sharding.kv_joint_table_ok(32, 1) and sharding.kv_joint_table_ok(64, 32)
    assert not sharding.kv_joint_table_ok(65, 1) and not sharding.kv_joint_table_ok(32, 16384)
    assert 2 * 32 * 64 * 4 <= sharding.SMALL_TABLE_BYTES < 32 * 512 * 4  # a 512-token prefill chunk keeps its layer-chunk pipeline


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype", [((6, 5, 8, 1, 128), "f16"), ((64, 8, 8, 1, 128), "bf16"), ((3, 64, 8, 2, 128), "f16"), ((130, 3, 8, 1, 128), "f16"),
                                         ((2, 1, 16, 1, 128), "bf16"), ((3, 5, 8, 16, 128), "f16"), ((2, 9, 3, 7, 128), "bf16")])
def test_absmax_of_a_decode_append_takes_one_workgroup_per_group_and_token(shape, dtype):
    """kvq_absmax_tokens on a slice of few tokens (a batch-sharded decode append: one; the default threshold is 16): `absmax_fewtokens_k` — one 256-thread
    workgroup per (group, token), a plain store — instead of rows / 8 one-wave workgroups that all atomicMax into the same
    word. Same table bit for bit as the tile walk (knob quant_few_tokens = 0) and as the oracle; `accumulate` keeps what the
    table holds; more than 16 tokens, or at most 8 rows, stay on the tile walk."""
    from efficient_llm_inference_amd import _lib
    from efficient_llm_inference_amd import kernels as K
    G, B, H, T, D = shape
    x = seeded_kv(shape, dtype, seed=5, dist="heavy")
    xt = to_torch(x, dtype)
    ref = O.absmax_tokens(x, odt(dtype))
    tables = {}
    for knob in (16, 0):
        _lib.set_tunable("quant_few_tokens", knob)
        try:
            _lib.kernel_log_clear()
            got = K.absmax_tokens(xt)
            torch.cuda.synchronize()
            log = _lib.kernel_log()
            tables[knob] = got
            assert np.array_equal(to_numpy(got).view(np.uint32), np.asarray(ref, dtype=np.float32).view(np.uint32)), (knob, log)
            assert log[0].startswith("absmax_fewtokens_k<" if knob and B * H > 8 else "quant_tile_k<"), (knob, log)
            # accumulate: the table already holds values (another chunk's / a larger one): the result is the max with them
            seeded = torch.full_like(got, 0.125)
            seeded[::2] = 1e9
            K.absmax_tokens(xt, seeded, accumulate=True)
            want = torch.maximum(got, torch.full_like(got, 0.125))
            want[::2] = 1e9
            assert torch.equal(seeded, want), knob
        finally:
            _lib.set_tunable("quant_few_tokens", 16)
    assert torch.equal(tables[0], tables[16])
    # past the threshold: the tile walk
    x3 = to_torch(seeded_kv((2, 4, 8, 17, 128), dtype, seed=6), dtype)
    _lib.kernel_log_clear()
    K.absmax_tokens(x3)
    assert _lib.kernel_log()[0].startswith("quant_tile_k<"), _lib.kernel_log()
