"""Batch sharding of the hot path across the GPUs of one node.

The path partitions by prompt / batch row: a prompt's KV cache, its scales and its eviction
state never leave the GPU that owns the prompt, so quantise / dequantise / trim / pool run with NO
data-path collective. What crosses xGMI is one ``all_reduce`` of a handful of run counters at the
end of a benchmark (RCCL = backend "nccl" on ROCm; "gloo" in the CPU tests).

One process per GPU (``torch.distributed.run``), rank r owns prompts ``r, r+W, r+2W, ...``.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.distributed as dist


def world() -> tuple:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


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
    if dist.get_backend() != "nccl":
        return torch.device("cpu")
    return device if device is not None else torch.device("cuda", torch.cuda.current_device())


def barrier() -> None:
    """Rendezvous of all ranks (no-op in a single process). bench.py brackets its timed region
    with barrier() + torch.cuda.synchronize() on both sides."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    """MAX of a host scalar over all ranks: the wall time of a step is the slowest rank's. One
    8-byte ``all_reduce(MAX)`` over xGMI (RCCL) / TCP (gloo); identical result on every rank."""
    _, ws = world()
    if ws == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_reduce_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_results(local: Dict[str, float], device=None) -> Dict[str, float]:
    """Combine per-rank ``benchmark_method`` dicts: token counts and cache MB are summed, elapsed
    time is the max over ranks (ranks run concurrently), tokens/sec = total tokens / max elapsed.
    ONE collective pair on two tiny tensors; identical result on every rank."""
    rank, ws = world()
    out = dict(local)
    if ws == 1:
        out["n_ranks"] = 1
        return out
    device = _reduce_device(device)
    est = local.get("est_kv_cache_mb_avg", float("nan"))
    has_est = 0.0 if est != est else 1.0
    sums = torch.tensor([float(local.get("total_new_tokens", 0)), (est if has_est else 0.0), has_est,
                         float(local.get("n_prompts", 0))], dtype=torch.float64, device=device)
    maxs = torch.tensor([float(local.get("elapsed_sec", 0.0)),
                         float(local.get("gpu_peak_mb") or 0.0)], dtype=torch.float64, device=device)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    dist.all_reduce(maxs, op=dist.ReduceOp.MAX)
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
