"""KV eviction on MI355X: ``trim_kv_sliding_window`` and ``chunk_summarize_kv`` over HIP.

Same names, arguments and return structure as the reference
(reference src/cache/implementations.py:124-140 and :295-346): they take and return the legacy
tuple ``tuple_L[(k, v)]`` of ``[B, H, T, D]`` tensors.

MI355X-first differences (results identical):
  * all 2L tensors of a call are processed by ONE kernel launch (their base pointers travel in
    the kernel-argument segment) into ONE output buffer; the returned tuple holds views of it.
  * ``trim_kv_sliding_window`` materialises the window (the reference returns views and lets the
    next ``torch.cat`` move the bytes); when ``T <= window_size`` the input objects are returned
    untouched, exactly like the reference (:135).
  * ``chunk_summarize_kv`` is a single pass (the reference zero-pads with ``torch.cat``, calls
    ``mean`` and concatenates again: three passes, :326-343).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from .. import _lib, kernels


def _flatten(past_key_values) -> List[torch.Tensor]:
    flat: List[torch.Tensor] = []
    for k, v in past_key_values:
        flat.append(k)
        flat.append(v)
    return flat


def _by_signature(tensors: List[torch.Tensor]) -> Dict[tuple, List[int]]:
    """indices grouped by (shape, strides, dtype, device): one launch per group"""
    groups: Dict[tuple, List[int]] = {}
    for i, t in enumerate(tensors):
        if t.dim() != 4:
            raise ValueError(f"kvq: KV tensors must be [B,H,T,D], got {tuple(t.shape)}")
        _lib.require_gpu(t, "past_key_values")
        groups.setdefault((tuple(t.shape), tuple(t.stride()), t.dtype, t.device), []).append(i)
    return groups


def _prep(t: torch.Tensor) -> torch.Tensor:
    return t if (t.size(-1) == 1 or t.stride(-1) == 1) else t.contiguous()


def trim_kv_sliding_window(past_key_values: tuple, window_size: int) -> tuple:
    """Keep only the last ``window_size`` tokens of every K and V (reference
    implementations.py:124-140)."""
    flat = [_prep(t) for t in _flatten(past_key_values)]
    out: List[torch.Tensor] = list(flat)
    for (shape, _strides, dtype, device), idx in _by_signature(flat).items():
        B, H, T, D = shape
        if not T > window_size:
            continue  # unchanged objects, as the reference
        W = int(window_size)
        for c0 in range(0, len(idx), 256):
            part = idx[c0:c0 + 256]
            buf = torch.empty(len(part), B, H, W, D, dtype=dtype, device=device)
            kernels.window_compact([flat[i] for i in part], buf, W)
            for j, i in enumerate(part):
                out[i] = buf[j]
    # pairs whose K and V were both untouched keep the caller's own objects
    orig = _flatten(past_key_values)
    res = []
    for l in range(len(out) // 2):
        k = orig[2 * l] if out[2 * l] is flat[2 * l] else out[2 * l]
        v = orig[2 * l + 1] if out[2 * l + 1] is flat[2 * l + 1] else out[2 * l + 1]
        res.append((k, v))
    return tuple(res)


def chunk_summarize_kv(past_key_values: tuple, chunk_size: int, keep_last: int) -> tuple:
    """Replace tokens older than the last ``keep_last`` by mean-pooled summaries of ``chunk_size``
    tokens each; the ragged last chunk is zero-padded, i.e. still divided by ``chunk_size``
    (reference implementations.py:295-346)."""
    flat = [_prep(t) for t in _flatten(past_key_values)]
    out: List[torch.Tensor] = list(flat)
    for (shape, _strides, dtype, device), idx in _by_signature(flat).items():
        B, H, T, D = shape
        keep = min(int(keep_last), T)
        if T - keep <= 0:
            continue  # nothing to compress: unchanged objects (reference :316-318)
        Tout = kernels.chunk_summary_len(T, chunk_size, keep_last)
        for c0 in range(0, len(idx), 256):
            part = idx[c0:c0 + 256]
            buf = torch.empty(len(part), B, H, Tout, D, dtype=dtype, device=device)
            kernels.chunk_meanpool([flat[i] for i in part], buf, int(chunk_size), int(keep_last))
            for j, i in enumerate(part):
                out[i] = buf[j]
    orig = _flatten(past_key_values)
    res = []
    for l in range(len(out) // 2):
        k = orig[2 * l] if out[2 * l] is flat[2 * l] else out[2 * l]
        v = orig[2 * l + 1] if out[2 * l + 1] is flat[2 * l + 1] else out[2 * l + 1]
        res.append((k, v))
    return tuple(res)
