"""KV-cache quantisation on MI355X: the reference's ``src/quantization/ops.py`` surface over HIP.

Same names, arguments, return orders and exceptions as the reference
(reference src/quantization/__init__.py:12-19): ``quantize_int8_per_tensor``,
``quantize_int4_per_tensor_packed``, ``dequantize_int8_per_tensor``,
``dequantize_int4_per_tensor_packed``, ``QuantizedLayerKV``, ``QuantizedKVCache``.

What changed underneath (MI355X-first, results identical):
  * the reference keeps six Python lists of per-token tensors per layer and issues 2*L*T
    dequantise launches + 2*L T-way ``torch.cat`` per decode step (ops.py:213-269, :345-355).
    Here a KV set is ONE persistent buffer ``[G, B, H, Tcap, Dq]`` (G = layers) plus an fp32 scale
    table ``[G, Tcap]``; prefill quantises all layers x all tokens in one launch per K/V set,
    a decode append is one launch per set, and ``to_past_key_values`` is one launch per set.
  * no host synchronisation: scales never leave the device (the reference calls
    ``float(scale)`` per slice, ops.py:87,117).
  * arithmetic lives in libkvq_hip.so only; CPU tensors raise (no fallback).
"""
from __future__ import annotations

import os

from typing import List, Optional, Sequence, Tuple

import torch

from .. import _lib, kernels
from ..kernels import QDTYPE, packed_dim

_OUT_DTYPES = (torch.float16, torch.bfloat16, torch.float32)


# ----------------------------------------------------------------------------- per-tensor API


def _as_rows(x: torch.Tensor) -> torch.Tensor:
    """View an arbitrary tensor as [1, B, H, 1, D] (one scale over everything) without copying
    when it is a [B,H,1,D] slice — the shape QuantizedLayerKV.append feeds (ops.py:178-179)."""
    if x.dim() == 4 and x.size(2) == 1:
        return x.unsqueeze(0)
    D = x.size(-1)
    return x.reshape(1, 1, -1, 1, D)


def _quantize_per_tensor(x: torch.Tensor, eps: float, kind: str):
    _lib.require_gpu(x, "x")
    if x.numel() == 0:
        raise _lib.KvqError("kvq: cannot quantise an empty tensor")  # reference: abs().max() raises too
    xx = x.reshape(1) if x.dim() == 0 else x
    x5 = _as_rows(xx)
    Dq = packed_dim(kind, x5.size(4))
    q = torch.empty(tuple(xx.shape[:-1]) + (Dq,), dtype=QDTYPE[kind], device=x.device)
    scales = torch.empty(1, 1, dtype=torch.float32, device=x.device)
    ws = torch.empty(1, dtype=torch.float32, device=x.device)
    kernels.quant_tokens(x5, q.view(1, x5.size(1), x5.size(2), 1, Dq), scales, ws, kind, eps)
    if x.dim() == 0 and kind == "int8":
        q = q.reshape(())
    # stored scale: already rounded to x.dtype by the kernel, so this cast is exact (ops.py:30,65)
    return q, scales.reshape(()).to(x.dtype)


def quantize_int8_per_tensor(x: torch.Tensor, eps: float = 1e-8) -> Tuple[torch.Tensor, torch.Tensor]:
    """Symmetric per-tensor INT8: ``q = clamp(round(x/scale), -127, 127)``, ``scale = max(max|x|/127, eps)``
    computed in fp32 and returned rounded to ``x.dtype`` (reference ops.py:10-30)."""
    return _quantize_per_tensor(x, eps, "int8")


def quantize_int4_per_tensor_packed(x: torch.Tensor, eps: float = 1e-8) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """Symmetric per-tensor INT4 in [-8, 7], two values per byte along the last dim (even index
    in the high nibble; an odd last dim is padded). Returns ``(packed, scale, orig_last_dim)``
    (reference ops.py:33-65)."""
    q, s = _quantize_per_tensor(x, eps, "int4")
    return q, s, int(x.size(-1))


def _scale_table(scale, device) -> torch.Tensor:
    if isinstance(scale, torch.Tensor):
        return scale.detach().to(device=device, dtype=torch.float32).reshape(1, 1)
    return torch.full((1, 1), float(scale), dtype=torch.float32, device=device)


def _check_out_dtype(out_dtype):
    if out_dtype not in _OUT_DTYPES:
        raise TypeError(f"kvq: out_dtype must be one of {_OUT_DTYPES}, got {out_dtype}")


def dequantize_int8_per_tensor(q: torch.Tensor, scale: torch.Tensor, out_dtype: torch.dtype) -> torch.Tensor:
    """``(float(q) * float(scale)).to(out_dtype)`` (reference ops.py:68-90). ``scale`` stays on
    the device: no ``float(scale)`` sync."""
    _lib.require_gpu(q, "q")
    _check_out_dtype(out_dtype)
    if q.dtype != torch.int8:
        raise TypeError("q must be int8")
    q = q.contiguous()
    out = torch.empty(q.shape, dtype=out_dtype, device=q.device)
    if q.numel() == 0:
        return out
    D = q.size(-1) if q.dim() else 1
    kernels.dequant_tokens(q.view(1, 1, -1, 1, D), _scale_table(scale, q.device), out.view(1, 1, -1, 1, D), "int8")
    return out


def dequantize_int4_per_tensor_packed(packed: torch.Tensor, scale: torch.Tensor, orig_last_dim: int,
                                      out_dtype: torch.dtype) -> torch.Tensor:
    """Unpack (high nibble first), subtract 8, scale, cast; the pad column of an odd
    ``orig_last_dim`` is dropped (reference ops.py:93-133)."""
    _lib.require_gpu(packed, "packed")
    _check_out_dtype(out_dtype)
    if packed.dtype != torch.uint8:
        raise TypeError("packed must be uint8")
    D = int(orig_last_dim)
    if packed.dim() < 1 or packed.size(-1) != (D + 1) // 2:
        raise ValueError(f"packed last dim {tuple(packed.shape)} does not match orig_last_dim {D}")
    packed = packed.contiguous()
    out = torch.empty(tuple(packed.shape[:-1]) + (D,), dtype=out_dtype, device=packed.device)
    if out.numel() == 0:
        return out
    kernels.dequant_tokens(packed.view(1, 1, -1, 1, packed.size(-1)), _scale_table(scale, packed.device),
                           out.view(1, 1, -1, 1, D), "int4")
    return out


# ----------------------------------------------------------------------------- persistent store


def _skewed_capacity(n: int) -> int:
    """The smallest capacity >= n that is 8 more than a multiple of 16 tokens. A store's (batch row, kv head) rows are
    ``Tcap * Dq`` bytes apart; the attention kernels' waves walk 64 of those rows in lockstep at the same relative offset, and the
    memory controller's channel choice repeats with the row stride: measured on decode attention over a Llama-3-8B batch-8
    store of 16,384 tokens (profiles/r04h_attn_store_capacity_skew.txt), a power-of-two Tcap (2 MiB INT8 rows) takes 39.5 us per
    layer call, Tcap = 16384 + 8 (rows 1 KiB further apart) 38.5, + 64 43.4, + 128 41.4 — every capacity that is 8 mod 16 tokens
    measured 38.5-38.7, every multiple of 64 beyond the power of two 39.6-43.4. Quantise and dequantise do not care
    (profiles/r04a_quant_stride_table.md)."""
    if os.environ.get("KVQ_STORE_CAP_EXACT"):  # measurement knob: the capacity as asked for (the sweep behind the numbers above)
        return n
    return (n + 7) // 16 * 16 + 8


class _KVStore:
    """One quantised KV set in HBM: ``q [G,B,H,Tcap,Dq]`` + stored scales ``[G,Tcap]`` (fp32 view
    of the input-dtype value). G groups never share scales (layer x K|V); all groups hold the
    same number of tokens when driven through QuantizedKVCache, but each keeps its own length so
    that ``layers[i].append`` keeps working."""

    def __init__(self, kind: str, n_groups: int, device, eps: float = 1e-8):
        self.kind = kind
        self.G = int(n_groups)
        self.device = device
        self.eps = eps
        self.lens: List[int] = [0] * self.G
        self.q: Optional[torch.Tensor] = None
        self.scales: Optional[torch.Tensor] = None
        self.ws: Optional[torch.Tensor] = None
        self.cap = 0
        self.B = self.H = self.D = None
        self.in_dtype: Optional[torch.dtype] = None
        self._want = 0  # capacity requested before the first append fixed B, H, D
        # persistent dequantised staging (scope row N1): [G,B,H,cap,D] in the compute dtype; tokens
        # [0, staged) of every group are already dequantised, so a decode step only adds the new one
        self.stage: Optional[torch.Tensor] = None
        self.staged = 0

    # -- allocation ---------------------------------------------------------------------------
    def reserve(self, T: int) -> None:
        """Grow capacity to >= T tokens (amortised doubling; existing tokens are preserved)."""
        if self.B is None or T <= self.cap:
            self._want = max(self._want, T)
            return
        new_cap = _skewed_capacity(max(T, 2 * self.cap, 16))
        Dq = packed_dim(self.kind, self.D)
        q = torch.empty(self.G, self.B, self.H, new_cap, Dq, dtype=QDTYPE[self.kind], device=self.device)
        sc = torch.zeros(self.G, new_cap, dtype=torch.float32, device=self.device)
        used = max(self.lens)
        if self.q is not None and used:
            q[:, :, :, :used] = self.q[:, :, :, :used]
            sc[:, :used] = self.scales[:, :used]
        self.q, self.scales, self.cap = q, sc, new_cap
        if self.stage is not None:  # keep what is already dequantised
            st = torch.empty(self.G, self.B, self.H, new_cap, self.D, dtype=self.stage.dtype, device=self.device)
            if self.staged:
                st[:, :, :, :self.staged] = self.stage[:, :, :, :self.staged]
            self.stage = st

    def _bind(self, x: torch.Tensor) -> None:
        B, H, _, D = x.shape
        if self.B is None:
            self.B, self.H, self.D, self.in_dtype = B, H, D, x.dtype
            self.reserve(max(self._want, 1))
        elif (B, H, D) != (self.B, self.H, self.D) or x.dtype != self.in_dtype:
            raise ValueError(
                f"kvq: KV slice {tuple(x.shape)} {x.dtype} does not match the cache "
                f"[B={self.B},H={self.H},D={self.D}] {self.in_dtype}")

    def _workspace(self, n: int) -> torch.Tensor:
        if self.ws is None or self.ws.numel() < n:
            self.ws = torch.empty(max(n, 1024), dtype=torch.float32, device=self.device)
        return self.ws

    # -- quantise -----------------------------------------------------------------------------
    def append(self, xs: Sequence[torch.Tensor], g0: int = 0) -> None:
        """Quantise ``xs`` (one [B,H,Tn,D] tensor per group g0..g0+len) and append Tn tokens to
        each of those groups: ONE launch. All of them must currently hold the same length."""
        if isinstance(xs, torch.Tensor):  # one [n,B,H,Tn,D] view (e.g. a window of the staging buffer)
            n = xs.size(0)
            first = xs[0]
        else:
            xs = list(xs)
            n = len(xs)
            first = xs[0]
        self._bind(first)
        Tn = first.size(2)
        if Tn == 0:
            return
        t0 = self.lens[g0]
        if any(self.lens[g0 + i] != t0 for i in range(n)):
            raise ValueError("kvq: groups appended together must hold the same number of tokens")
        self.reserve(t0 + Tn)
        qwin = self.q[g0:g0 + n, :, :, t0:t0 + Tn]
        swin = self.scales[g0:g0 + n, t0:t0 + Tn]
        kernels.quant_tokens(xs, qwin, swin, self._workspace(n * Tn), self.kind, self.eps)
        for i in range(n):
            self.lens[g0 + i] = t0 + Tn

    # -- dequantise ---------------------------------------------------------------------------
    def dequant(self, out_dtype: torch.dtype, g0: int = 0, n: Optional[int] = None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Dequantise groups g0..g0+n (all tokens) into ``[n,B,H,T,D]``: ONE launch."""
        n = self.G - g0 if n is None else n
        T = self.lens[g0]
        if T == 0 or self.q is None:
            raise ValueError("Empty cache")  # reference ops.py:219-220
        if any(self.lens[g0 + i] != T for i in range(n)):
            raise ValueError("kvq: groups dequantised together must hold the same number of tokens")
        if out is None:
            out = torch.empty(n, self.B, self.H, T, self.D, dtype=out_dtype, device=self.device)
        kernels.dequant_tokens(self.q[g0:g0 + n, :, :, :T], self.scales[g0:g0 + n, :T], out, self.kind)
        return out

    def dequant_staged(self, out_dtype: torch.dtype) -> torch.Tensor:
        """All groups, all tokens, as a VIEW ``[G,B,H,T,D]`` of the persistent staging buffer:
        only tokens appended since the last call are dequantised (one launch, O(new tokens)
        instead of the reference's O(T) re-dequantise + T-way cat per step, ops.py:213-269).
        Values are identical to :meth:`dequant`. Earlier views stay valid: staged tokens are
        never rewritten."""
        T = self.lens[0]
        if T == 0 or self.q is None:
            raise ValueError("Empty cache")
        if any(n != T for n in self.lens):
            raise ValueError("kvq: groups dequantised together must hold the same number of tokens")
        if self.stage is None or self.stage.dtype != out_dtype:
            self.stage = torch.empty(self.G, self.B, self.H, self.cap, self.D, dtype=out_dtype, device=self.device)
            self.staged = 0
        if self.staged < T:
            s0 = self.staged
            kernels.dequant_tokens(self.q[:, :, :, s0:T], self.scales[:, s0:T], self.stage[:, :, :, s0:T], self.kind)
            self.staged = T
        return self.stage[:, :, :, :T]

    # -- accounting ---------------------------------------------------------------------------
    def stored_bytes(self, g: int) -> int:
        """numel*itemsize of stored slices + scales, scale itemsize = input dtype's
        (what the reference's lists would hold: ops.py:271-290)."""
        if self.B is None:
            return 0
        T = self.lens[g]
        return T * (self.B * self.H * packed_dim(self.kind, self.D) + self.in_dtype.itemsize)


_MODE_KINDS = {"int8": ("int8", "int8"), "int4": ("int4", "int4"), "mixed": ("int8", "int4")}


class QuantizedLayerKV:
    """Quantised KV cache of one transformer layer (reference ops.py:136-290).

    Modes: ``"int8"`` (K, V int8), ``"int4"`` (both packed int4), ``"mixed"`` (K int8, V int4 —
    reference ops.py:201-210).
    """

    def __init__(self, mode: str = "int8", device: str = "cuda", compute_dtype: torch.dtype = torch.float16,
                 _stores: Optional[Tuple[_KVStore, _KVStore]] = None, _group: int = 0):
        assert mode in ["int8", "int4", "mixed"], f"Invalid mode: {mode}"
        self.mode = mode
        self.device = device
        self.compute_dtype = compute_dtype
        kk, vk = _MODE_KINDS[mode]
        if _stores is None:
            _stores = (_KVStore(kk, 1, device), _KVStore(vk, 1, device))
        self._k, self._v = _stores
        self._g = _group

    def __len__(self) -> int:
        return self._k.lens[self._g]

    @torch.no_grad()
    def append(self, k_1tok: torch.Tensor, v_1tok: torch.Tensor) -> None:
        """Append key/value slices ``[B, H, 1, D]`` (any number of tokens ``[B,H,n,D]`` is accepted:
        each token still gets its own scale, as n single-token appends would)."""
        self._k.append([k_1tok], self._g)
        self._v.append([v_1tok], self._g)

    @torch.no_grad()
    def get_kv(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Full dequantised ``(K, V)``, each ``[B, H, T, D]`` in ``compute_dtype``."""
        if len(self) == 0:
            raise ValueError("Empty cache")
        k = self._k.dequant(self.compute_dtype, self._g, 1)[0]
        v = self._v.dequant(self.compute_dtype, self._g, 1)[0]
        return k, v

    def estimated_bytes(self) -> int:
        """Bytes of the quantised stores + scales (reference ops.py:271-290)."""
        return self._k.stored_bytes(self._g) + self._v.stored_bytes(self._g)

    # -- the reference's list attributes, materialised on demand as views ------------------------
    def _slices(self, st: _KVStore):
        T = st.lens[self._g]
        return [st.q[self._g, :, :, t:t + 1] for t in range(T)] if st.q is not None else []

    def _scale_list(self, st: _KVStore):
        T = st.lens[self._g]
        if st.scales is None:
            return []
        s = st.scales[self._g, :T].to(st.in_dtype)
        return [s[t] for t in range(T)]

    @property
    def k_store(self):
        return self._slices(self._k)

    @property
    def v_store(self):
        return self._slices(self._v)

    @property
    def k_scales(self):
        return self._scale_list(self._k)

    @property
    def v_scales(self):
        return self._scale_list(self._v)

    @property
    def k_meta(self):
        return [self._k.D if self._k.kind == "int4" else None] * len(self)

    @property
    def v_meta(self):
        return [self._v.D if self._v.kind == "int4" else None] * len(self)


class QuantizedKVCache:
    """Multi-layer quantised KV cache (reference ops.py:293-363): all layers of K share one HBM
    buffer and one launch, likewise V."""

    def __init__(self, n_layers: int, mode: str = "int8", device: str = "cuda",
                 compute_dtype: torch.dtype = torch.float16, incremental: bool = True):
        assert mode in ["int8", "int4", "mixed"], f"Invalid mode: {mode}"
        self.incremental = incremental
        self.mode = mode
        self.device = device
        self.compute_dtype = compute_dtype
        kk, vk = _MODE_KINDS[mode]
        self._k = _KVStore(kk, n_layers, device)
        self._v = _KVStore(vk, n_layers, device)
        self.layers = [
            QuantizedLayerKV(mode=mode, device=device, compute_dtype=compute_dtype,
                             _stores=(self._k, self._v), _group=i)
            for i in range(n_layers)
        ]

    def reserve(self, n_tokens: int) -> None:
        """Pre-size the stores (e.g. prompt + max_new_tokens) so decode never reallocates."""
        self._k.reserve(n_tokens)
        self._v.reserve(n_tokens)

    def _check_layers(self, past_key_values) -> None:
        if len(past_key_values) != len(self.layers):
            raise ValueError(f"kvq: got {len(past_key_values)} layers, cache has {len(self.layers)}")

    @torch.no_grad()
    def append_from_past(self, past_key_values: tuple) -> None:
        """Quantise and append the LAST token of every layer's ``(k, v)`` (reference ops.py:323-330):
        two launches (K set, V set) instead of 2L."""
        self._check_layers(past_key_values)
        self._k.append([k[:, :, -1:, :] for k, _ in past_key_values])
        self._v.append([v[:, :, -1:, :] for _, v in past_key_values])

    @torch.no_grad()
    def init_from_prompt_past(self, past_key_values: tuple) -> None:
        """Quantise every prompt token of every layer, one scale per (layer, K|V, token)
        (reference ops.py:333-342 loops L*T times): two launches."""
        self._check_layers(past_key_values)
        self._k.append([k for k, _ in past_key_values])
        self._v.append([v for _, v in past_key_values])

    @torch.no_grad()
    def to_past_key_values(self, copy: bool = False) -> tuple:
        """Dequantise to the legacy tuple ``tuple_L[(K, V)]``, each ``[B,H,T,D]`` in compute_dtype
        (reference ops.py:345-355): two launches. With ``incremental=True`` (default) the tuple
        holds VIEWS of two persistent staging buffers and only tokens appended since the previous
        call are dequantised — read-only by contract: writing into them corrupts every later step,
        which the reference's fresh tensors (ops.py:267-269) would not. ``copy=True`` (or
        ``incremental=False`` on the cache) re-dequantises everything into fresh buffers the caller
        owns, as the reference does. Same values either way."""
        if not self.layers:
            return tuple()
        if not copy and self.incremental and len(set(self._k.lens)) == 1 and len(set(self._v.lens)) == 1:
            # persistent staging: dequantise only what was appended since the last call
            k = self._k.dequant_staged(self.compute_dtype)
            v = self._v.dequant_staged(self.compute_dtype)
        else:
            k = self._k.dequant(self.compute_dtype)
            v = self._v.dequant(self.compute_dtype)
        return tuple((k[i], v[i]) for i in range(len(self.layers)))

    @torch.no_grad()
    def attend(self, layer: int, q: torch.Tensor, k_new: Optional[torch.Tensor] = None,
               v_new: Optional[torch.Tensor] = None, sm_scale: Optional[float] = None,
               append: bool = False) -> torch.Tensor:
        """Single-token attention of ``layer`` straight from the quantised store (no dequantised copy):
        ``softmax(q K^T * sm_scale) V`` over every stored token, plus the exact ``k_new`` / ``v_new``
        ``[B, Hkv, D]`` as one more token when given — what ``to_past_key_values()`` followed by the
        model's attention computes (reference ops.py:345-355, benchmarker.py:470-471), within fp16
        tolerance. ``q`` is ``[B, Hq, D]`` fp16 / bf16 (Hq a multiple of the cache's kv heads). With
        ``append=True`` the new token is also quantised into the store (``append_from_past`` for this
        layer, ops.py:323-330) by the same call."""
        k, v = self._k, self._v
        T = k.lens[layer]
        if T != v.lens[layer] or k.q is None:
            raise ValueError("Empty cache")
        if append and (k_new is None or v_new is None):
            raise ValueError("kvq: append=True needs k_new and v_new")
        if q.dim() != 3:
            raise ValueError(f"kvq: attend takes a [B, Hq, D] query, got {tuple(q.shape)}")
        B, Hq, D = q.shape
        if append:
            # an appended token must be what append_from_past would have been given: the store's own
            # input dtype (the scale is rounded to it, estimated_bytes counts its itemsize) and shape
            for name, t in (("k_new", k_new), ("v_new", v_new)):
                if t.dtype != k.in_dtype or tuple(t.shape) != (k.B, k.H, k.D):
                    raise ValueError(f"kvq: attend(append=True) {name} {tuple(t.shape)} {t.dtype} does not match the cache "
                                     f"[B={k.B},H={k.H},D={k.D}] {k.in_dtype}")
            k.reserve(T + 1)
            v.reserve(T + 1)
        scale = float(sm_scale) if sm_scale is not None else D ** -0.5
        out = torch.empty_like(q)
        # the workspace size (a host loop over the capacity) and the per-layer plans are looked up only when
        # the shape, the capacity or the store allocation changes — not once per call
        ws_key = (B, Hq, D, q.dtype, k.cap, q.device)
        if getattr(self, "_attn_ws_key", None) != ws_key:
            need = kernels.decode_attn_workspace_cap(B, Hq, k.H, max(k.cap, T + 1), D)
            if getattr(self, "_attn_ws", None) is None or self._attn_ws.numel() < need or self._attn_ws.device != q.device:
                self._attn_ws = torch.empty(need, dtype=torch.float32, device=q.device)
            self._attn_ws_key = ws_key
            self._attn_plans = {}
        if append:
            pkey = (k.q.data_ptr(), v.q.data_ptr(), q.dtype, B, Hq, D)
            plan = self._attn_plans.get(layer)
            if plan is None or plan.key[:2] != (k.q[layer].data_ptr(), v.q[layer].data_ptr()) or plan.key[2:] != pkey[2:]:
                plan = self._attn_plans[layer] = kernels.DecodeStepPlan(q, k.q[layer], k.scales[layer], k.kind, v.q[layer],
                                                                        v.scales[layer], v.kind, k.eps)
            kernels.decode_step(plan, q, k_new, v_new, T, out, self._attn_ws, scale)
            k.lens[layer] = v.lens[layer] = T + 1
        else:
            kernels.decode_attn(q, k.q[layer], k.scales[layer], k.kind, v.q[layer], v.scales[layer], v.kind, T, out,
                                self._attn_ws, scale, k_new, v_new)
        return out

    def estimated_bytes(self) -> int:
        """Total bytes over layers (reference ops.py:357-363)."""
        return sum(layer.estimated_bytes() for layer in self.layers)
