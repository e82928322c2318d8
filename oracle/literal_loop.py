"""The reference's CPU call STRUCTURE, restated with torch-CPU ops: one tiny op chain per
``[B,H,1,D]`` slice and a T-way ``torch.cat`` — what ``QuantizedLayerKV.get_kv`` does on a host
without the CUDA plugin (reference src/quantization/ops.py:213-269 with the fallback branches
:88-90 and :120-133), and ``init_from_prompt_past`` for the quantise side (:333-342, :10-65).

TEST INFRASTRUCTURE ONLY (see kvq_oracle.py): used by bench.py's ``cpu_baseline`` leg as the
"literal" variant BASELINE.md §4 asks for, next to the vectorised C port. Parity: checked against
the numpy oracle in tests/test_oracle_c.py.
"""
from __future__ import annotations

import torch


def quantize_slices(x: torch.Tensor, kind: str, eps: float = 1e-8):
    """x [B,H,T,D] -> (list of T quantised slices, list of T 0-dim scales): the per-token loop of
    init_from_prompt_past (ops.py:339-342) over quantize_int8_per_tensor / _int4_..._packed."""
    qmax, qmin = (127.0, -127) if kind == "int8" else (7.0, -8)
    qs, scales = [], []
    for t in range(x.size(2)):
        sl = x[:, :, t:t + 1, :]
        x32 = sl.float()
        scale = (x32.abs().max() / qmax).clamp(min=eps)
        q = torch.clamp((x32 / scale).round(), qmin, qmax).to(torch.int8)
        if kind == "int4":
            if q.size(-1) % 2 == 1:
                q = torch.cat([q, torch.zeros_like(q[..., :1])], dim=-1)
            u = (q + 8).to(torch.uint8)
            q = (u[..., 0::2] << 4) | u[..., 1::2]
        qs.append(q)
        scales.append(scale.to(sl.dtype))
    return qs, scales


def dequantize_slices(qs, scales, kind: str, D: int, out_dtype: torch.dtype) -> torch.Tensor:
    """list of T slices -> [B,H,T,D]: the per-slice dequantise + T-way cat of get_kv (ops.py:222-268)."""
    outs = []
    for q, s in zip(qs, scales):
        if kind == "int4":
            hi = (q >> 4) & 0x0F
            lo = q & 0x0F
            u = torch.empty((*q.shape[:-1], q.shape[-1] * 2), dtype=torch.uint8)
            u[..., 0::2] = hi
            u[..., 1::2] = lo
            q = (u.to(torch.int16) - 8).to(torch.int8)[..., :D]
        outs.append((q.float() * s.float()).to(out_dtype))
    return torch.cat(outs, dim=2)
