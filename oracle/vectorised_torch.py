"""The reference's CPU ARITHMETIC as whole-tensor torch-CPU ops: every token slice of a
``[G,B,H,T,D]`` KV set quantised / dequantised at once (one ``amax`` over (B,H,D) per token instead
of one op chain per slice). Same arithmetic as quantize_int8_per_tensor / _int4_per_tensor_packed /
dequantize_* (reference src/quantization/ops.py:10-65, :88-90, :120-133) without the reference's
per-slice Python loop — the fair multi-core CPU ceiling BASELINE.md §4 calls "vectorised".

TEST INFRASTRUCTURE ONLY (see kvq_oracle.py): used by bench.py's ``cpu_baseline`` leg. Parity:
checked bit-for-bit against the numpy oracle in tests/test_oracle_c.py.
"""
from __future__ import annotations

import torch


def quantize_tokens(x: torch.Tensor, kind: str, eps: float = 1e-8):
    """x [G,B,H,T,D] -> (q [G,B,H,T,Dq], stored scales [G,T] in x.dtype)."""
    qmax, qmin = (127.0, -127.0) if kind == "int8" else (7.0, -8.0)
    x32 = x.float()                                                          # ops.py:26 / :47
    s32 = (x32.abs().amax(dim=(1, 2, 4)) / qmax).clamp(min=eps)              # [G,T]  ops.py:27-28 / :48-49
    q = torch.clamp((x32 / s32[:, None, None, :, None]).round(), qmin, qmax).to(torch.int8)  # ops.py:29 / :50
    if kind == "int4":
        if q.size(-1) % 2 == 1:
            q = torch.cat([q, torch.zeros_like(q[..., :1])], dim=-1)         # ops.py:54-56
        u = (q + 8).to(torch.uint8)                                          # ops.py:59
        q = (u[..., 0::2] << 4) | u[..., 1::2]                               # ops.py:61-63
    return q, s32.to(x.dtype)                                                # ops.py:30 / :65


def dequantize_tokens(q: torch.Tensor, scales: torch.Tensor, kind: str, D: int, out_dtype: torch.dtype) -> torch.Tensor:
    """q [G,B,H,T,Dq], stored scales [G,T] -> [G,B,H,T,D] of out_dtype."""
    if kind == "int4":
        u = torch.empty((*q.shape[:-1], q.shape[-1] * 2), dtype=torch.uint8)
        u[..., 0::2] = (q >> 4) & 0x0F                                       # ops.py:122-127
        u[..., 1::2] = q & 0x0F
        q = (u.to(torch.int16) - 8).to(torch.int8)[..., :D]                  # ops.py:129-131
    return (q.float() * scales.float()[:, None, None, :, None]).to(out_dtype)  # ops.py:90 / :133


def trim_kv_sliding_window(x: torch.Tensor, window_size: int) -> torch.Tensor:
    """x [..., T, D] -> the last ``window_size`` tokens, materialised (the reference returns the view
    ``k[:, :, -W:, :]`` and pays the copy in HF's next ``cat``; src/cache/implementations.py:124-140)."""
    T = x.size(-2)
    return x if T <= window_size else x[..., T - window_size:, :].contiguous()


def chunk_summarize_kv(x: torch.Tensor, chunk_size: int, keep_last: int) -> torch.Tensor:
    """x [B,H,T,D]: the reference's own op chain (src/cache/implementations.py:313-345): zero-pad the old tokens to
    a multiple of ``chunk_size``, ``view(B,H,n,chunk,D).mean(dim=3)``, ``cat`` with the recent tail."""
    B, H, T, D = x.shape
    keep = min(keep_last, T)
    old_len = T - keep
    if old_len <= 0:
        return x
    old, recent = x[:, :, :old_len, :], x[:, :, old_len:, :]
    pad = (chunk_size - old_len % chunk_size) % chunk_size
    if pad:
        old = torch.cat([old, torch.zeros(B, H, pad, D, dtype=x.dtype)], dim=2)   # :326-333
    n = old.size(2) // chunk_size
    summ = old.view(B, H, n, chunk_size, D).mean(dim=3)                              # :338-339
    return torch.cat([summ, recent], dim=2)                                          # :344
