"""Batch sharding of the hot path across the GPUs of one node.

The path partitions by prompt / batch row: a prompt's KV cache, its scales and its eviction
state never leave the GPU that owns the prompt, so quantise / dequantise / trim / pool run with NO
data-path collective. What crosses xGMI is

* one ``all_reduce`` pair over a handful of run counters at the end of a benchmark, and
* — only when ONE genuinely batched ``[B>1,H,1,D]`` slice is split by batch rows over ranks — one
  ``all_reduce(MAX)`` of the ``[G,T]`` fp32 abs-max table between the abs-max and the quantise
  phases (`quantize_tokens_batch_sharded`): the reference's scale spans the whole batch
  (``abs().max()`` of the slice, reference src/quantization/ops.py:27,48), so every rank must
  quantise with the same scale.

One process per GPU, rank r owns prompts ``r, r+W, r+2W, ...``. RCCL = backend "nccl" on ROCm;
"gloo" in the CPU tests. `init_distributed` brings both up: a gloo group as the control plane (it
cannot fail on xGMI / IPC trouble) and an RCCL group for every reduction; whether RCCL came up is
AGREED over gloo, so the ranks either all use it or all fail (or, with ``allow_gloo``, all fall
back) — never a mix that would hang at the next collective.
"""
from __future__ import annotations

import contextlib
import datetime
import os
import sys
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

# the group + device every reduction of this module uses; None = torch's default group
_STATE = {"group": None, "device": None, "backend": None, "local": False}


@contextlib.contextmanager
def _stdout_to_stderr():
    """File descriptor 1 points at stderr for the duration: gloo / RCCL print connection chatter ("[Gloo] Rank 0 is
    connected to ...") to STDOUT from C++ while a group comes up, and a benchmark's stdout is its one JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


class RcclUnavailable(RuntimeError):
    """RCCL could not be brought up on every rank and the caller did not allow the gloo fallback."""


def init_distributed(rank: int, world_size: int, device: Optional[torch.device], *, allow_gloo: bool = False,
                     ranks_share_device: bool = False, timeout_s: float = 300.0) -> str:
    """Bring up the process groups of an N-rank run and return the backend the reductions use
    ("nccl" = RCCL over xGMI, or "gloo").

    MASTER_ADDR / MASTER_PORT come from the environment (127.0.0.1 on one node). ``device`` None
    = a CPU-only run (tests, launcher self-test): gloo only. ``ranks_share_device``: more ranks
    than GPUs (a rehearsal on a small box) — RCCL refuses duplicate devices, so it is not tried.
    RCCL failing on ANY rank raises `RcclUnavailable` on EVERY rank unless ``allow_gloo``."""
    timeout = datetime.timedelta(seconds=timeout_s)
    with _stdout_to_stderr():
        dist.init_process_group("gloo", rank=rank, world_size=world_size, timeout=timeout)
        probe = torch.zeros(1)
        dist.all_reduce(probe)  # gloo connects its pairs lazily: do it (and its chatter) here
    _STATE.update(group=None, device=None, backend="gloo")
    if device is None or device.type != "cuda":
        return "gloo"
    group, why = None, ""
    if ranks_share_device:
        why = "several ranks share one GPU (RCCL refuses duplicate devices)"
    else:
        # two attempts, every rank in lockstep (new_group is itself a collective over the gloo group): eager
        # communicator creation bound to this rank's device first, torch's lazy initialisation second
        for kwargs in ({"device_id": device}, {}):
            ok, g = 1, None
            try:
                with _stdout_to_stderr():
                    g = dist.new_group(backend="nccl", timeout=timeout, **kwargs)
                    warm = torch.zeros(1, device=device)
                    dist.all_reduce(warm, group=g)  # RCCL initialises lazily: surface its errors here
                    torch.cuda.synchronize(device)
            except Exception as exc:  # noqa: BLE001 - any RCCL / IPC / driver failure
                ok, why = 0, (why + "; " if why else "") + repr(exc)
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # over gloo: every rank learns the same answer
            if int(flag.item()) == 1:
                group = g
                break
            if g is not None and ok:  # it worked here but not everywhere: drop it and stay in lockstep
                try:
                    dist.destroy_process_group(g)
                except Exception:  # noqa: BLE001
                    pass
    if group is not None:
        _STATE.update(group=group, device=device, backend="nccl")
        return "nccl"
    if not allow_gloo:
        msg = (f"rank {rank}: RCCL is not usable on every rank"
               + (f" (this rank: {why})" if why else " (it failed on another rank)")
               + "; pass --allow-gloo-timing to run the timing reduction over gloo instead")
        dist.destroy_process_group()
        raise RcclUnavailable(msg)
    return "gloo"


def shutdown() -> None:
    """Barrier + destroy every group this module created (idempotent)."""
    if dist.is_available() and dist.is_initialized():
        try:
            dist.barrier()
        finally:
            if _STATE["group"] is not None:
                try:
                    dist.destroy_process_group(_STATE["group"])
                except Exception:  # noqa: BLE001
                    pass
            _STATE.update(group=None, device=None, backend=None)
            dist.destroy_process_group()


def backend() -> Optional[str]:
    """"nccl" | "gloo" | None (single process)"""
    if _STATE["local"] or not (dist.is_available() and dist.is_initialized()):
        return None
    return _STATE["backend"] or dist.get_backend()


def world() -> tuple:
    if _STATE["local"]:
        return 0, 1
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


@contextlib.contextmanager
def local_mode():
    """Inside the block this rank behaves as a single process: `world()` is (0, 1), `barrier()` / `max_over_ranks()` /
    `aggregate_results()` touch no collective. For per-rank side measurements of an N-rank run that must not be able to
    hang the job: a rank that fails inside the block fails alone (bench.py's sub-records at N > 1)."""
    prev = _STATE["local"]
    _STATE["local"] = True
    try:
        yield
    finally:
        _STATE["local"] = prev


def shard_prompts(prompts: Sequence, rank: int = None, world_size: int = None) -> List:
    """Round-robin shard: rank r takes prompts r, r+W, ... (balanced within one prompt)."""
    if rank is None or world_size is None:
        rank, world_size = world()
    return list(prompts[rank::world_size])


def shard_batch_rows(n_rows: int, rank: int = None, world_size: int = None) -> range:
    """Contiguous block of batch rows for this rank (config 5: batch 64 over 8 GPUs = 8 rows
    each). Rows of one rank are contiguous so its KV slab ``[L,2,rows,H,T,D]`` is one allocation."""
    if rank is None or world_size is None:
        rank, world_size = world()
    base, extra = divmod(n_rows, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def _reduce_device(device=None):
    """where the scalars of a reduction live: the GPU under RCCL, host memory under gloo"""
    if backend() != "nccl":
        return torch.device("cpu")
    if _STATE["device"] is not None:
        return _STATE["device"]
    return device if device is not None else torch.device("cuda", torch.cuda.current_device())


def _all_reduce(t: torch.Tensor, op) -> None:
    dist.all_reduce(t, op=op, group=_STATE["group"])


def barrier() -> None:
    """Rendezvous of all ranks (no-op in a single process). bench.py brackets its timed region
    with barrier() + torch.cuda.synchronize() on both sides."""
    if not _STATE["local"] and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if _STATE["group"] is not None:  # RCCL: a 1-element all_reduce on the device, then drained
            t = torch.zeros(1, device=_STATE["device"])
            _all_reduce(t, dist.ReduceOp.SUM)
            torch.cuda.synchronize(_STATE["device"])
        else:
            dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    """MAX of a host scalar over all ranks: the wall time of a step is the slowest rank's. One
    8-byte ``all_reduce(MAX)`` over xGMI (RCCL) / TCP (gloo); identical result on every rank."""
    _, ws = world()
    if ws == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_reduce_device(device))
    _all_reduce(t, dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_results(local: Dict[str, float], device=None) -> Dict[str, float]:
    """Combine per-rank ``benchmark_method`` dicts: token and prompt counts are summed, elapsed time
    is the max over ranks (ranks run concurrently), tokens/sec = total tokens / max elapsed, and
    ``est_kv_cache_mb_avg`` stays what the single-process dict reports — the mean over PROMPTS
    (reference benchmarker.py:804-809) — i.e. per-rank means weighted by their finite-estimate
    counts (``n_est``; ``n_prompts`` when the caller does not supply it).
    ONE collective pair on two tiny tensors; identical result on every rank."""
    rank, ws = world()
    out = dict(local)
    if ws == 1:
        out["n_ranks"] = 1
        return out
    device = _reduce_device(device)
    est = local.get("est_kv_cache_mb_avg", float("nan"))
    n_est = float(local.get("n_est", local.get("n_prompts", 0))) if est == est else 0.0
    sums = torch.tensor([float(local.get("total_new_tokens", 0)), (est * n_est if n_est else 0.0), n_est,
                         float(local.get("n_prompts", 0))], dtype=torch.float64, device=device)
    maxs = torch.tensor([float(local.get("elapsed_sec", 0.0)),
                         float(local.get("gpu_peak_mb") or 0.0)], dtype=torch.float64, device=device)
    _all_reduce(sums, dist.ReduceOp.SUM)
    _all_reduce(maxs, dist.ReduceOp.MAX)
    total_tokens, est_sum, est_n, n_prompts = sums.tolist()
    elapsed, peak = maxs.tolist()
    out.update({
        "total_new_tokens": int(total_tokens),
        "elapsed_sec": elapsed,
        "tokens_per_sec": total_tokens / elapsed if elapsed > 0 else float("inf"),
        "est_kv_cache_mb_avg": est_sum / est_n if est_n > 0 else float("nan"),
        "gpu_peak_mb": peak if peak > 0 else None,
        "n_prompts": int(n_prompts),
        "n_ranks": ws,
    })
    return out


def benchmark_sharded(benchmarker, prompts: Sequence[str], method: str, **kw) -> Dict[str, float]:
    """``benchmark_method`` over this rank's share of ``prompts`` + aggregation."""
    mine = shard_prompts(prompts)
    res = benchmarker.benchmark_method(mine, method=method, **kw) if mine else {
        "method": method, "elapsed_sec": 0.0, "total_new_tokens": 0, "tokens_per_sec": 0.0,
        "est_kv_cache_mb_avg": float("nan"), "gpu_peak_mb": None}
    res["n_prompts"] = len(mine)
    return aggregate_results(res)


# --------------------------------------------------------------------------- the one exchange step

def all_reduce_absmax(table: torch.Tensor, force: bool = False) -> torch.Tensor:
    """``all_reduce(MAX)`` of the ``[G,T]`` fp32 abs-max table IN PLACE: the single data-path
    collective of the path. RCCL reduces the device tensor directly over xGMI
    (≤ 8 MiB at config 5; 256 B per decode step); under gloo (CPU tests, or a 1-GPU rehearsal)
    the table makes a round trip through host memory. MAX of non-negative floats is exact and
    order-independent, so every rank ends up with bit-identical scales."""
    _, ws = world()
    if ws == 1 and not force:
        return table
    if table.dtype != torch.float32:
        raise TypeError("abs-max table must be fp32")
    if table.is_cuda and backend() != "nccl":
        host = table.detach().cpu()
        _all_reduce(host, dist.ReduceOp.MAX)
        table.copy_(host)
    else:
        _all_reduce(table, dist.ReduceOp.MAX)
    return table


# Layer chunks exist to put chunk i's all_reduce(MAX) under chunk i + 1's abs-max pass; each chunk costs three launches
# (abs-max, collective, quantise). Round 3 first sized them to the 256 MiB Infinity Cache (<= 64 MiB of local input, so
# that the quantise phase's re-read would be served from it): measured on one MI355X the re-read runs at HBM speed all
# the same (19.0 us for the INT8 phase of a 64 MiB chunk = 5.3 TB/s of its 100.9 MB; profiles/r03p_*), and the 64
# short launches per set cost more than they save — so: ONE chunk on a single rank (nothing to overlap), CHUNKS_PER_SET
# chunks when a collective has to hide. KVQ_SHARD_CHUNK_BYTES overrides (A-B runs).
CHUNKS_PER_SET = 4
SMALL_TABLE_BYTES = int(os.environ.get("KVQ_SHARD_SMALL_TABLE_BYTES", str(16 << 10)))  # (0: per set and per layer chunk, as round 3 — A-B runs); [G,T] fp32 tables up to this size are exchanged whole (and K + V together: quantize_kv_batch_sharded)
CHUNK_BYTES = int(os.environ.get("KVQ_SHARD_CHUNK_BYTES", "0"))  # 0 = by rank count (above)


class ShardedQuantBuffers:
    """Persistent outputs + scratch of :func:`quantize_tokens_batch_sharded` for one KV set: the quantised rows
    ``q [G,B_local,H,T,Dq]``, the stored scales ``[G,T]``, the abs-max table ``[G,T]`` (one slice per layer chunk is
    what crosses the ranks), the layer-chunk plan and — on more than one rank — the side stream and events that put
    chunk i's ``all_reduce(MAX)`` under chunk i + 1's abs-max pass. Built once per (shape, kind); reused every step."""

    def __init__(self, x_local, kind: str, chunk_bytes: int = None, force_overlap: bool = False):
        from . import kernels as K
        first = x_local[0]
        G = len(x_local)
        B, H, T, D = first.shape
        dev = first.device
        self.kind, self.shape = kind, (G, B, H, T, D)
        self.q = torch.empty(G, B, H, T, K.packed_dim(kind, D), dtype=K.QDTYPE[kind], device=dev)
        self.scales = torch.empty(G, T, dtype=torch.float32, device=dev)
        self.absmax = torch.empty(G, T, dtype=torch.float32, device=dev)
        per_group = max(1, B * H * T * D * first.element_size())
        _, ws = world()
        if chunk_bytes is None:
            chunk_bytes = CHUNK_BYTES
        if chunk_bytes > 0:
            self.groups_per_chunk = max(1, min(G, chunk_bytes // per_group))
        elif G * T * 4 <= SMALL_TABLE_BYTES and not force_overlap:
            # a decode append (T = 1) or a few tokens: the [G,T] table is a few hundred bytes, its all_reduce is pure latency
            # and there is no abs-max pass long enough to hide it under — ONE chunk, ONE collective
            self.groups_per_chunk = G
        else:
            self.groups_per_chunk = G if (ws == 1 and not force_overlap) else max(1, -(-G // CHUNKS_PER_SET))
        self.chunks = [(g0, min(G, g0 + self.groups_per_chunk)) for g0 in range(0, G, self.groups_per_chunk)]
        self.n_chunks = len(self.chunks)
        # force_overlap: take the side-stream path on ONE rank too (a 1-GPU box's only way to exercise it under RCCL)
        self.overlap = (ws > 1 or force_overlap) and dev.type == "cuda" and backend() == "nccl"
        if self.overlap:
            self.comm_stream = torch.cuda.Stream(device=dev)
            self.ev_absmax = [torch.cuda.Event() for _ in self.chunks]
            self.ev_reduced = [torch.cuda.Event() for _ in self.chunks]


def quantize_tokens_batch_sharded(x_local, kind: str, eps: float = 1e-8, out: "ShardedQuantBuffers" = None, two_phase: bool = None):
    """Quantise this rank's batch rows ``x_local`` (``[G,B_local,H,T,D]`` on the GPU, or a list of G
    ``[B_local,H,T,D]`` tensors) of a slice whose batch is split over the ranks, with the scale of
    the WHOLE batch (reference ops.py:27,48: one ``abs().max()`` per ``[B,H,1,D]`` slice), layer chunk by layer chunk:

        abs-max of the chunk's local rows (HIP)
        -> all_reduce(MAX) of the chunk's [Gc,T] table (RCCL on a side stream, under the NEXT chunk's abs-max pass)
        -> quantise the chunk with those abs-max values (HIP; the rows are read a SECOND time, at HBM speed: measured,
           chunks sized to the 256 MiB Infinity Cache did not make the re-read cheaper — see CHUNKS_PER_SET above)

    The two passes are the floor of a slice whose scale crosses ranks: 1.67 x (INT8) / 1.80 x (INT4) the single pass's
    HBM traffic, each pass at 0.76-0.79 of the 8 TB/s peak on its own bytes (profiles/traffic.json).

    Returns ``(q, scales)`` as `kernels.quant_tokens` does for the un-sharded batch: ``q`` holds
    this rank's rows, ``scales`` ``[G,T]`` (stored scales widened to fp32) is identical on every
    rank and bit-identical to the un-sharded result. ``out``: reuse these buffers (a decode / prefill loop builds them
    once); the returned tensors are then ``out.q`` / ``out.scales``.

    On a single rank there is nothing to exchange and the slice takes the un-sharded call (`kernels.quant_tokens`: ONE pass
    over the input — the 1024-thread register tile for batches of up to 131072 elements per token); ``two_phase=True``
    runs the phases a rank of a larger job runs anyway (bench.py reports both at N = 1)."""
    from . import kernels as K
    if out is None:
        out = ShardedQuantBuffers(x_local, kind)
    G, B, H, T, D = out.shape
    if len(x_local) != G or tuple(x_local[0].shape) != (B, H, T, D) or out.kind != kind:
        raise ValueError(f"kvq: ShardedQuantBuffers were built for {out.shape} {out.kind}, got {len(x_local)} x {tuple(x_local[0].shape)} {kind}")
    _, ws = world()
    if two_phase is None:
        two_phase = ws > 1 or out.overlap
    if not two_phase:
        if ws > 1:
            raise ValueError("kvq: a batch split over ranks needs the abs-max exchange (two_phase=False is for one rank)")
        K.quant_tokens(x_local, out.q, out.scales, out.absmax, kind, eps)
        return out.q, out.scales

    def quant(c):
        g0, g1 = out.chunks[c]
        K.quant_tokens_with_absmax(x_local[g0:g1], out.absmax[g0:g1], kind, eps, q=out.q[g0:g1], scales=out.scales[g0:g1])

    pending = None  # chunk whose table is being reduced while the next chunk's abs-max runs
    out.absmax.zero_()  # ONE fill for the whole pass; every chunk's abs-max launch accumulates into its slice
    for c, (g0, g1) in enumerate(out.chunks):
        K.absmax_tokens(x_local[g0:g1], out.absmax[g0:g1], accumulate=True)
        if ws == 1 and not out.overlap:
            quant(c)
            continue
        if out.overlap:
            main = torch.cuda.current_stream(out.q.device)
            out.ev_absmax[c].record(main)
            with torch.cuda.stream(out.comm_stream):
                out.comm_stream.wait_event(out.ev_absmax[c])
                all_reduce_absmax(out.absmax[g0:g1], force=True)
                out.ev_reduced[c].record(out.comm_stream)
            if pending is not None:
                main.wait_event(out.ev_reduced[pending])
                quant(pending)
            pending = c
        else:  # gloo (CPU tests, 1-GPU rehearsal): the table makes its round trip through host memory in stream order
            all_reduce_absmax(out.absmax[g0:g1])
            quant(c)
    if pending is not None:
        torch.cuda.current_stream(out.q.device).wait_event(out.ev_reduced[pending])
        quant(pending)
    return out.q, out.scales


def kv_joint_table_ok(G: int, T: int) -> bool:
    """Does a K + V pair of G-tensor sets with T tokens take the joint path of :func:`quantize_kv_batch_sharded`?
    (2G tensors fit the 128-entry pointer table of ONE launch; the [2G,T] fp32 table is small enough that its exchange is latency.)"""
    return 2 * G <= 128 and 2 * G * T * 4 <= SMALL_TABLE_BYTES


def quantize_kv_batch_sharded(k_local, v_local, kinds, eps: float = 1e-8, outs=None, two_phase: bool = None):
    """One decode step's (or a short chunk's) K AND V slices of a batch that is split over the ranks, together:

        ONE abs-max launch over the 2G tensors of both sets -> ONE all_reduce(MAX) of the [2G,T] table -> two quantise launches

    instead of two abs-max launches and two collectives (`quantize_tokens_batch_sharded` per set, which on several ranks also
    cuts each set into layer chunks to hide the collective under the next chunk's abs-max pass — right for a prefill chunk,
    wrong for a decode append: its table is `2 * L * 4` bytes, its collectives are pure latency, and fewer of them is the
    only lever: xGMI all_reduce latency, not bandwidth). Taken when the joint table is at most SMALL_TABLE_BYTES; larger
    slices fall back to the per-set pipeline. Returns ``((qk, k_scales), (qv, v_scales))``, bit-identical to the per-set calls
    and to the un-sharded quantise (reference ops.py:27,48: one scale per [B,H,1,D] slice of the WHOLE batch).
    ``outs``: a pair of ShardedQuantBuffers to reuse."""
    from . import kernels as K
    if outs is None:
        outs = (ShardedQuantBuffers(k_local, kinds[0]), ShardedQuantBuffers(v_local, kinds[1]))
    ok, ov = outs
    G, B, H, T, D = ok.shape
    _, ws = world()
    if two_phase is None:
        two_phase = ws > 1
    k0, v0 = k_local[0], v_local[0]
    joint = (two_phase and kv_joint_table_ok(G, T) and ov.shape == ok.shape and len(k_local) == G and len(v_local) == G
             and k0.dtype == v0.dtype and k0.stride() == v0.stride())  # one launch = one dtype and one stride triple
    if not joint:
        return (quantize_tokens_batch_sharded(k_local, kinds[0], eps, ok, two_phase),
                quantize_tokens_batch_sharded(v_local, kinds[1], eps, ov, two_phase))
    if getattr(ok, "joint", None) is None or ok.joint.shape != (2 * G, T):
        ok.joint = torch.empty(2 * G, T, dtype=torch.float32, device=ok.q.device)
    if isinstance(k_local, torch.Tensor) and isinstance(v_local, torch.Tensor) and k_local.stride() == v_local.stride():
        # 2G separately addressed [B,H,T,D] tensors, one launch; the pointer table is arithmetic on the two bases and is
        # kept while the caller hands in the same two buffers (a serving loop's staging tensors)
        both = getattr(ok, "joint_in", None)
        if both is None or both.key != K.TensorGroups.key_of((k_local, v_local)):
            both = ok.joint_in = K.TensorGroups((k_local, v_local))
    else:
        both = list(k_local) + list(v_local)  # the legacy-tuple form (per-layer tensors)
    K.absmax_tokens(both, ok.joint)
    all_reduce_absmax(ok.joint)
    K.quant_tokens_with_absmax(k_local, ok.joint[:G], kinds[0], eps, q=ok.q, scales=ok.scales)
    K.quant_tokens_with_absmax(v_local, ok.joint[G:], kinds[1], eps, q=ov.q, scales=ov.scales)
    return (ok.q, ok.scales), (ov.q, ov.scales)
