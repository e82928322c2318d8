"""Tensor-level launchers over the C ABI (include/kvq_hip.h).

Every function takes torch tensors that already live on the GPU, validates shapes on the host
(a wrong shape must never reach a hand-written kernel), and enqueues ONE asynchronous launch
on torch's current stream. Nothing here synchronises, allocates device memory behind the
caller's back (except where a docstring says it returns a new tensor), or falls back to torch ops.

KV sets are 5-D views ``[G, B, H, T, D]`` (G = independent scale groups: layer x K|V), or a
list of G separately allocated ``[B, H, T, D]`` tensors (the reference's legacy tuple layout,
reference src/quantization/ops.py:178-179).
"""
from __future__ import annotations

from typing import Sequence, Union

import torch

from . import _lib
from ._lib import KvqStrides, byref, c_void_p, check, dims5, dtype_code, require_gpu, strides4

def _launch(dev, fn, *args):
    """``fn(*args, stream)`` on ``dev``'s current torch stream with ``dev`` made the calling thread's current device for
    the call: the library launches on the CURRENT device (and refuses a buffer that lives elsewhere: KVQ_E_DEVICE), so a
    single-process multi-GPU caller may hand over tensors of any device."""
    with _lib.device_guard(dev):
        return fn(*args, _lib.current_stream(dev))


KIND_BITS = {"int8": 8, "int4": 4}
QDTYPE = {"int8": torch.int8, "int4": torch.uint8}

TensorOrList = Union[torch.Tensor, Sequence[torch.Tensor]]


def packed_dim(kind: str, D: int) -> int:
    """last dim of the quantised store: D int8 values or ceil(D/2) packed bytes (ops.py:54-63)"""
    return D if kind == "int8" else (D + 1) // 2


class TensorGroups:
    """The G separately addressed ``[B,H,T,D]`` tensors of one launch with the pointer table already built: the
    concatenation of stacked ``[Gi,B,H,T,D]`` tensors that share shape, dtype and strides (e.g. a decode step's K and V
    sets, `sharding.quantize_kv_batch_sharded`) without 2G Python views and ``data_ptr()`` calls per step. Accepted
    wherever a tensor list is; holds the tensors alive."""

    def __init__(self, stacked: Sequence[torch.Tensor]):
        first = stacked[0]
        for t in stacked:
            if t.dim() != 5:
                raise _lib.KvqError(f"kvq: expected [G,B,H,T,D] tensors, got shape {tuple(t.shape)}")
            require_gpu(t, "input")
            if t.shape[1:] != first.shape[1:] or t.dtype != first.dtype or t.stride() != first.stride() or t.device != first.device:
                raise _lib.KvqError("kvq: all tensors of one launch must share shape, dtype, strides and device")
        if first.size(4) > 1 and first.stride(4) != 1:
            raise _lib.KvqError("kvq: last dim must be contiguous")
        step = first.stride(0) * first.element_size()
        ptrs = [t.data_ptr() + g * step for t in stacked for g in range(t.size(0))]
        if len(ptrs) > 256:
            raise _lib.KvqError("kvq: more than 256 tensors in one call")
        _, sb, sh, st, _ = first.stride()
        self.arr, self.strides = _lib.ptr_array(ptrs), KvqStrides(0, sb, sh, st)
        self.dims, self.dtype, self.device, self.keep = (len(ptrs),) + tuple(first.shape[1:]), first.dtype, first.device, list(stacked)
        self.key = tuple((t.data_ptr(), t.size(0)) for t in stacked) + (first.stride(), first.shape[1:], first.dtype)

    @staticmethod
    def key_of(stacked):
        first = stacked[0]
        return tuple((t.data_ptr(), t.size(0)) for t in stacked) + (first.stride(), first.shape[1:], first.dtype)


def _in_views(x: TensorOrList):
    """-> (base_ptr|None, ptr_array|None, strides, (G,B,H,T,D), dtype, device, keepalive)"""
    if isinstance(x, TensorGroups):
        return None, x.arr, x.strides, x.dims, x.dtype, x.device, x
    if isinstance(x, torch.Tensor):
        if x.dim() != 5:
            raise _lib.KvqError(f"kvq: expected [G,B,H,T,D], got shape {tuple(x.shape)}")
        require_gpu(x, "input")
        return c_void_p(x.data_ptr()), None, strides4(x), tuple(x.shape), x.dtype, x.device, x
    xs = list(x)
    if not xs:
        raise _lib.KvqError("kvq: empty tensor list")
    first = xs[0]
    for t in xs:
        if t.dim() != 4:
            raise _lib.KvqError(f"kvq: expected a list of [B,H,T,D] tensors, got shape {tuple(t.shape)}")
        require_gpu(t, "input")
        if t.shape != first.shape or t.dtype != first.dtype or t.stride() != first.stride() or t.device != first.device:
            # same shape, dtype AND strides: one stride triple describes every group of a launch
            raise _lib.KvqError("kvq: all tensors of one launch must share shape, dtype, strides and device")
    if first.size(3) > 1 and first.stride(3) != 1:
        raise _lib.KvqError("kvq: last dim must be contiguous")
    if len(xs) > 256:
        raise _lib.KvqError("kvq: more than 256 tensors in one call")
    sb, sh, st, _ = first.stride()
    B, H, T, D = first.shape
    arr = _lib.ptr_array([t.data_ptr() for t in xs])
    return None, arr, KvqStrides(0, sb, sh, st), (len(xs), B, H, T, D), first.dtype, first.device, xs


def quant_tokens(x: TensorOrList, q: torch.Tensor, scales: torch.Tensor, absmax_ws: torch.Tensor,
                 kind: str, eps: float = 1e-8) -> None:
    """Per-token symmetric quantise of ``x`` into the store window ``q`` and the scale table.

    x       [G,B,H,T,D] (or list of G [B,H,T,D]) fp16 / bf16 / fp32
    q       [G,B,H,T,Dq] int8 (kind="int8") or uint8 packed (kind="int4"); may be a T-window of
            a larger [G,B,H,Tcap,Dq] store
    scales  [G,T] fp32 window of the scale table (row stride arbitrary): receives the stored
            scale (rounded to x's dtype) widened to fp32
    absmax_ws fp32 workspace with >= G*T elements
    Reference: quantize_int8_per_tensor / quantize_int4_per_tensor_packed per [B,H,1,D] slice
    (ops.py:10-65 under ops.py:174-210, :333-342).
    """
    bits = KIND_BITS[kind]
    base, arr, ist, (G, B, H, T, D), dt, dev, _keep = _in_views(x)
    require_gpu(q, "q")
    require_gpu(scales, "scales")
    require_gpu(absmax_ws, "absmax_ws")
    Dq = packed_dim(kind, D)
    if tuple(q.shape) != (G, B, H, T, Dq) or q.dtype != QDTYPE[kind]:
        raise _lib.KvqError(f"kvq: store window must be {(G, B, H, T, Dq)} {QDTYPE[kind]}, got {tuple(q.shape)} {q.dtype}")
    if tuple(scales.shape) != (G, T) or scales.dtype != torch.float32 or (T > 1 and scales.stride(1) != 1):
        raise _lib.KvqError(f"kvq: scales window must be fp32 {(G, T)} with unit token stride")
    if absmax_ws.dtype != torch.float32 or absmax_ws.numel() < G * T or not absmax_ws.is_contiguous():
        raise _lib.KvqError("kvq: absmax_ws must be a contiguous fp32 tensor with >= G*T elements")
    if G * B * H * T * D == 0:
        return
    lib = _lib.load()
    fn = lib.kvq_quant_i8_tokens if bits == 8 else lib.kvq_quant_i4_tokens
    rc = _launch(dev, fn, base, arr, byref(ist), dtype_code(dt), c_void_p(q.data_ptr()), byref(strides4(q)),
            c_void_p(scales.data_ptr()), scales.stride(0), c_void_p(absmax_ws.data_ptr()), float(eps),
            byref(dims5(G, B, H, T, D)))
    check(rc, f"quant_tokens[{kind}]")


def absmax_tokens(x: TensorOrList, out: torch.Tensor = None, accumulate: bool = False) -> torch.Tensor:
    """Phase 1 of the sharded-batch quantise (kvq_absmax_tokens): ``[G,T]`` fp32 table of
    ``max |x[g, :, :, t, :]|`` over THIS rank's batch rows; returns ``out`` (allocated when None).
    ``accumulate``: ``out`` already holds non-negative values and the result is the max with them
    (kvq_absmax_tokens_acc: no fill launch; the caller zeroed the table, e.g. once for a layer-chunked pass).
    Reference: the ``x32.abs().max()`` of quantize_int8 / int4_per_tensor (ops.py:27,48) before it is
    completed across ranks by ``sharding.all_reduce_absmax``."""
    base, arr, ist, (G, B, H, T, D), dt, dev, _keep = _in_views(x)
    if out is None:
        if accumulate:
            raise _lib.KvqError("kvq: absmax_tokens(accumulate=True) needs the table to accumulate into")
        out = torch.empty(G, T, dtype=torch.float32, device=dev)
    require_gpu(out, "absmax")
    if tuple(out.shape) != (G, T) or out.dtype != torch.float32 or not out.is_contiguous():
        raise _lib.KvqError(f"kvq: absmax table must be contiguous fp32 {(G, T)}")
    lib = _lib.load()
    fn = lib.kvq_absmax_tokens_acc if accumulate else lib.kvq_absmax_tokens
    check(_launch(dev, fn, base, arr, byref(ist), dtype_code(dt), c_void_p(out.data_ptr()), byref(dims5(G, B, H, T, D))),
          "absmax_tokens")
    return out


def quant_tokens_with_absmax(x: TensorOrList, absmax: torch.Tensor, kind: str, eps: float = 1e-8,
                             q: torch.Tensor = None, scales: torch.Tensor = None):
    """Phase 2 (kvq_quant_tokens_from_absmax): quantise ``x`` with the scale
    ``max(absmax[g,t] / QMAX, eps)`` — ``absmax`` being the table of :func:`absmax_tokens`, completed
    across ranks when the batch is sharded. Returns ``(q [G,B,H,T,Dq], scales [G,T])`` (allocated
    when not given); with the local table on one rank this equals :func:`quant_tokens` bit for bit."""
    bits = KIND_BITS[kind]
    base, arr, ist, (G, B, H, T, D), dt, dev, _keep = _in_views(x)
    Dq = packed_dim(kind, D)
    if q is None:
        q = torch.empty(G, B, H, T, Dq, dtype=QDTYPE[kind], device=dev)
    if scales is None:
        scales = torch.empty(G, T, dtype=torch.float32, device=dev)
    for name, t in (("q", q), ("scales", scales), ("absmax", absmax)):
        require_gpu(t, name)
    if tuple(q.shape) != (G, B, H, T, Dq) or q.dtype != QDTYPE[kind]:
        raise _lib.KvqError(f"kvq: store window must be {(G, B, H, T, Dq)} {QDTYPE[kind]}, got {tuple(q.shape)} {q.dtype}")
    if tuple(scales.shape) != (G, T) or scales.dtype != torch.float32 or (T > 1 and scales.stride(1) != 1):
        raise _lib.KvqError(f"kvq: scales window must be fp32 {(G, T)} with unit token stride")
    if tuple(absmax.shape) != (G, T) or absmax.dtype != torch.float32 or not absmax.is_contiguous():
        raise _lib.KvqError(f"kvq: absmax table must be contiguous fp32 {(G, T)}")
    if G * B * H * T * D:
        check(_launch(dev, _lib.load().kvq_quant_tokens_from_absmax, 
            bits, base, arr, byref(ist), dtype_code(dt), c_void_p(q.data_ptr()), byref(strides4(q)),
            c_void_p(scales.data_ptr()), scales.stride(0), c_void_p(absmax.data_ptr()), float(eps),
            byref(dims5(G, B, H, T, D))), f"quant_tokens_with_absmax[{kind}]")
    return q, scales


def dequant_tokens(q: torch.Tensor, scales: torch.Tensor, out: torch.Tensor, kind: str) -> None:
    """``out[g,b,h,t,:] = RN(float(q[g,b,h,t,:]) * scales[g,t])`` for a whole KV set, one launch.

    q [G,B,H,T,Dq], scales [G,T] fp32, out [G,B,H,T,D] fp16/bf16/fp32 (windows of larger
    buffers allowed). Reference: dequantize_*_per_tensor per slice + T-way torch.cat
    (ops.py:68-133, :213-269).
    """
    bits = KIND_BITS[kind]
    for t, n in ((q, "q"), (scales, "scales"), (out, "out")):
        require_gpu(t, n)
    if q.dim() != 5 or out.dim() != 5:
        raise _lib.KvqError("kvq: q and out must be 5-D [G,B,H,T,D]")
    G, B, H, T, D = out.shape
    Dq = packed_dim(kind, D)
    if tuple(q.shape) != (G, B, H, T, Dq) or q.dtype != QDTYPE[kind]:
        raise _lib.KvqError(f"kvq: q must be {(G, B, H, T, Dq)} {QDTYPE[kind]}, got {tuple(q.shape)} {q.dtype}")
    if tuple(scales.shape) != (G, T) or scales.dtype != torch.float32 or (T > 1 and scales.stride(1) != 1):
        raise _lib.KvqError(f"kvq: scales must be fp32 {(G, T)} with unit token stride")
    lib = _lib.load()
    if G * B * H * T * D == 0:
        return
    fn = lib.kvq_dequant_i8_tokens if bits == 8 else lib.kvq_dequant_i4_tokens
    rc = _launch(out.device, fn, c_void_p(q.data_ptr()), byref(strides4(q)), c_void_p(scales.data_ptr()), scales.stride(0),
            c_void_p(out.data_ptr()), byref(strides4(out)), dtype_code(out.dtype),
            byref(dims5(G, B, H, T, D)))
    check(rc, f"dequant_tokens[{kind}]")


def dequant_i8_flat(q: torch.Tensor, scale: float) -> torch.Tensor:
    """Reference entry ``kvq_ext.dequant_int8_to_fp16(q, scale)`` (extensions.py:70-86):
    returns a new fp16 tensor of q's shape. q must be contiguous int8 on the GPU."""
    require_gpu(q, "q")
    if q.dtype != torch.int8:
        raise _lib.KvqError("q must be int8")
    if not q.is_contiguous():
        raise _lib.KvqError("q must be contiguous")
    out = torch.empty(q.shape, dtype=torch.float16, device=q.device)
    rc = _launch(q.device, _lib.load().kvq_dequant_i8_f16_flat, c_void_p(q.data_ptr()), float(scale), c_void_p(out.data_ptr()),
                                             q.numel())
    check(rc, "dequant_i8_flat")
    return out


def dequant_i4_flat(packed: torch.Tensor, scale: float, orig_last_dim: int) -> torch.Tensor:
    """Reference entry ``kvq_ext.dequant_int4_packed_to_fp16(packed, scale, orig_last_dim)``
    (extensions.py:88-114): new fp16 tensor, last dim = 2*packed_last, pad column zeroed."""
    require_gpu(packed, "packed")
    if packed.dtype != torch.uint8:
        raise _lib.KvqError("packed must be uint8")
    if not packed.is_contiguous():
        raise _lib.KvqError("packed must be contiguous")
    if packed.dim() < 1:
        raise _lib.KvqError("packed must have at least 1 dim")
    sizes = list(packed.shape)
    packed_last = sizes[-1]
    sizes[-1] = packed_last * 2
    out = torch.empty(sizes, dtype=torch.float16, device=packed.device)
    rc = _launch(packed.device, _lib.load().kvq_dequant_i4_f16_flat, c_void_p(packed.data_ptr()), float(scale), c_void_p(out.data_ptr()),
                                             packed.numel(), packed_last, int(orig_last_dim))
    check(rc, "dequant_i4_flat")
    return out


def window_compact(x: TensorOrList, out: torch.Tensor, window: int) -> None:
    """``out[g,b,h,0:W,:] = x[g,b,h,T-W:T,:]``, W = min(window, T). out: [G,B,H,W,D], same dtype.
    Reference: trim_kv_sliding_window (src/cache/implementations.py:124-140), materialised."""
    base, arr, ist, (G, B, H, T, D), dt, dev, _keep = _in_views(x)
    require_gpu(out, "out")
    W = min(int(window), T)
    if window < 0 or tuple(out.shape) != (G, B, H, W, D) or out.dtype != dt:
        raise _lib.KvqError(f"kvq: out must be {(G, B, H, W, D)} {dt}, got {tuple(out.shape)} {out.dtype}")
    if dt.itemsize not in (2, 4):
        raise _lib.KvqError(f"kvq: unsupported element size {dt.itemsize}")
    if G * B * H * W * D == 0:
        return
    rc = _launch(dev, _lib.load().kvq_window_compact, base, arr, byref(ist), c_void_p(out.data_ptr()), byref(strides4(out)),
                                        dt.itemsize, int(window), byref(dims5(G, B, H, T, D)))
    check(rc, "window_compact")


def chunk_summary_len(T: int, chunk_size: int, keep_last: int) -> int:
    """output length of chunk_summarize_kv (implementations.py:313-345); pure host arithmetic"""
    keep = min(int(keep_last), int(T))
    old = T - keep
    if old <= 0:
        return int(T)
    return (old + chunk_size - 1) // chunk_size + keep


def chunk_meanpool(x: TensorOrList, out: torch.Tensor, chunk_size: int, keep_last: int) -> None:
    """Chunk-summary mean-pool of the old tokens + exact copy of the last ``keep_last``.
    out: [G,B,H,chunk_summary_len(T),D], same dtype. Reference: chunk_summarize_kv
    (implementations.py:295-346)."""
    base, arr, ist, (G, B, H, T, D), dt, dev, _keep = _in_views(x)
    require_gpu(out, "out")
    if chunk_size <= 0 or keep_last < 0:
        raise _lib.KvqError("kvq: chunk_size must be > 0 and keep_last >= 0")
    Tout = chunk_summary_len(T, chunk_size, keep_last)
    if tuple(out.shape) != (G, B, H, Tout, D) or out.dtype != dt:
        raise _lib.KvqError(f"kvq: out must be {(G, B, H, Tout, D)} {dt}, got {tuple(out.shape)} {out.dtype}")
    if G * B * H * T * D == 0:
        return
    rc = _launch(dev, _lib.load().kvq_chunk_meanpool, base, arr, byref(ist), c_void_p(out.data_ptr()), byref(strides4(out)),
                                        dtype_code(dt), int(chunk_size), int(keep_last),
                                        byref(dims5(G, B, H, T, D)))
    check(rc, "chunk_meanpool")


def gather_tokens(x: TensorOrList, out: torch.Tensor, idx: torch.Tensor) -> None:
    """``out[g,b,h,j,:] = x[g,b,h,idx[j],:]``; idx: int32 DEVICE tensor [n]. out: [G,B,H,n,D], same
    dtype. Reference: ``index_select(2, idx)`` + ``torch.cat`` of the sparse eviction family
    (src/cache/implementations.py:143-292)."""
    base, arr, ist, (G, B, H, T, D), dt, dev, _keep = _in_views(x)
    require_gpu(out, "out")
    require_gpu(idx, "idx")
    n = idx.numel()
    if idx.dtype != torch.int32 or idx.dim() != 1 or not idx.is_contiguous():
        raise _lib.KvqError("kvq: idx must be a contiguous 1-D int32 tensor")
    if tuple(out.shape) != (G, B, H, n, D) or out.dtype != dt:
        raise _lib.KvqError(f"kvq: out must be {(G, B, H, n, D)} {dt}, got {tuple(out.shape)} {out.dtype}")
    if dt.itemsize not in (2, 4):
        raise _lib.KvqError(f"kvq: unsupported element size {dt.itemsize}")
    if G * B * H * n * D == 0:
        return
    rc = _launch(dev, _lib.load().kvq_gather_tokens, base, arr, byref(ist), c_void_p(out.data_ptr()), byref(strides4(out)),
                                       dt.itemsize, c_void_p(idx.data_ptr()), n, byref(dims5(G, B, H, T, D)))
    check(rc, "gather_tokens")


def decode_attn_workspace(B: int, Hq: int, Hkv: int, T: int, D: int) -> int:
    """fp32 elements of scratch :func:`decode_attn` needs for these dims (include/kvq_hip.h)."""
    n = int(_lib.load().kvq_decode_attn_workspace(byref(_lib.KvqAttnDims(B, Hq, Hkv, T, D))))
    if n < 0:
        raise _lib.KvqError(f"kvq: bad decode-attention dims B={B} Hq={Hq} Hkv={Hkv} T={T} D={D}")
    return n


def decode_attn_workspace_cap(B: int, Hq: int, Hkv: int, Tcap: int, D: int) -> int:
    """fp32 elements that cover :func:`decode_attn` / :func:`decode_step` for EVERY context length up to
    ``Tcap`` (the per-T size is not monotone): what a decode loop allocates once."""
    n = int(_lib.load().kvq_decode_attn_workspace_cap(byref(_lib.KvqAttnDims(B, Hq, Hkv, Tcap, D))))
    if n < 0:
        raise _lib.KvqError(f"kvq: bad decode-attention dims B={B} Hq={Hq} Hkv={Hkv} T={Tcap} D={D}")
    return n


def decode_attn(q: torch.Tensor, k_store: torch.Tensor, k_scales: torch.Tensor, k_kind: str,
                v_store: torch.Tensor, v_scales: torch.Tensor, v_kind: str, T: int, out: torch.Tensor,
                workspace: torch.Tensor, sm_scale: float, k_new: torch.Tensor | None = None,
                v_new: torch.Tensor | None = None) -> None:
    """One decode step of ONE layer straight from the quantised store (kvq_decode_attn).

    q / out ``[B, Hq, D]``, k_new / v_new ``[B, Hkv, D]`` (fp16 or bf16, last dim contiguous, any
    batch / head strides); k_store / v_store ``[B, Hkv, Tcap, Dq]`` views of one layer of the store
    (int8, or uint8 packed INT4); k_scales / v_scales ``[>= T]`` fp32 rows of the scale table. The
    first ``T`` stored tokens are attended, plus the exact new token when given. Replaces
    ``to_past_key_values`` + the model's attention over the dequantised past
    (reference src/quantization/ops.py:345-355, src/benchmarking/benchmarker.py:470-471).
    """
    for name, t in (("q", q), ("out", out), ("k_store", k_store), ("v_store", v_store), ("k_scales", k_scales),
                    ("v_scales", v_scales), ("workspace", workspace)):
        require_gpu(t, name)
    if q.dim() != 3 or out.shape != q.shape or out.dtype != q.dtype:
        raise _lib.KvqError(f"kvq: decode_attn q / out must be matching [B,Hq,D] tensors, got {tuple(q.shape)} / {tuple(out.shape)}")
    if q.dtype not in (torch.float16, torch.bfloat16):
        raise _lib.KvqError(f"kvq: decode_attn computes from fp16 / bf16 queries, got {q.dtype}")
    B, Hq, D = q.shape
    if k_store.dim() != 4 or v_store.dim() != 4 or k_store.shape[:2] != v_store.shape[:2] or k_store.size(0) != B:
        raise _lib.KvqError("kvq: decode_attn stores must be [B,Hkv,Tcap,Dq] views of one layer")
    Hkv = k_store.size(1)
    if k_store.dtype != QDTYPE[k_kind] or v_store.dtype != QDTYPE[v_kind]:
        raise _lib.KvqError("kvq: decode_attn store dtype does not match its kind")
    if k_store.size(3) != packed_dim(k_kind, D) or v_store.size(3) != packed_dim(v_kind, D):
        raise _lib.KvqError("kvq: decode_attn store last dim does not match head_dim")
    T = int(T)
    if T < 0 or T > k_store.size(2) or T > v_store.size(2) or k_scales.numel() < T or v_scales.numel() < T:
        raise _lib.KvqError(f"kvq: decode_attn T={T} exceeds the store / scale table")
    for name, t in (("q", q), ("out", out), ("k_store", k_store), ("v_store", v_store)):
        if t.stride(-1) != 1:
            raise _lib.KvqError(f"kvq: decode_attn {name} last dim must be contiguous")
    if k_scales.dtype != torch.float32 or v_scales.dtype != torch.float32 or workspace.dtype != torch.float32:
        raise _lib.KvqError("kvq: decode_attn scales / workspace must be float32")
    if (T > 0 and (k_scales.stride(-1) != 1 or v_scales.stride(-1) != 1)) or not workspace.is_contiguous():
        raise _lib.KvqError("kvq: decode_attn scales / workspace must be contiguous")
    if (k_new is None) != (v_new is None):
        raise _lib.KvqError("kvq: decode_attn k_new and v_new go together")
    kn = vn = (None, 0, 0)
    if k_new is not None:
        for name, t in (("k_new", k_new), ("v_new", v_new)):
            require_gpu(t, name)
            if t.shape != (B, Hkv, D) or t.dtype != q.dtype or t.stride(-1) != 1:
                raise _lib.KvqError(f"kvq: decode_attn {name} must be [B,Hkv,D] of the query dtype")
        kn = (c_void_p(k_new.data_ptr()), k_new.stride(0), k_new.stride(1))
        vn = (c_void_p(v_new.data_ptr()), v_new.stride(0), v_new.stride(1))
    kst = KvqStrides(0, k_store.stride(0), k_store.stride(1), k_store.stride(2))  # 1-byte elements: bytes
    vst = KvqStrides(0, v_store.stride(0), v_store.stride(1), v_store.stride(2))
    dims = _lib.KvqAttnDims(B, Hq, Hkv, T, D)
    check(_launch(q.device, _lib.load().kvq_decode_attn, 
        c_void_p(q.data_ptr()), q.stride(0), q.stride(1),
        c_void_p(k_store.data_ptr()), byref(kst), c_void_p(k_scales.data_ptr()), KIND_BITS[k_kind],
        c_void_p(v_store.data_ptr()), byref(vst), c_void_p(v_scales.data_ptr()), KIND_BITS[v_kind],
        kn[0], kn[1], kn[2], vn[0], vn[1], vn[2],
        c_void_p(out.data_ptr()), out.stride(0), out.stride(1), dtype_code(q.dtype), float(sm_scale),
        c_void_p(workspace.data_ptr()), workspace.numel(), byref(dims)), "decode_attn")


class DecodeStepPlan:
    """Pre-validated arguments of :func:`decode_step` for one layer of one store: everything that
    does not change from token to token (pointers, strides, kinds), so that the per-step host cost
    is one ctypes call. Rebuild it when the store is reallocated or the query layout changes."""

    __slots__ = ("key", "k_ptr", "v_ptr", "ks_ptr", "vs_ptr", "kst", "vst", "kbits", "vbits", "B", "Hq", "Hkv", "D",
                 "cap", "dtype_code", "eps", "keep")

    def __init__(self, q: torch.Tensor, k_store: torch.Tensor, k_scales: torch.Tensor, k_kind: str,
                 v_store: torch.Tensor, v_scales: torch.Tensor, v_kind: str, eps: float):
        for name, t in (("q", q), ("k_store", k_store), ("v_store", v_store), ("k_scales", k_scales), ("v_scales", v_scales)):
            require_gpu(t, name)
        if q.dim() != 3 or q.dtype not in (torch.float16, torch.bfloat16):
            raise _lib.KvqError(f"kvq: decode_step needs a [B,Hq,D] fp16 / bf16 query, got {tuple(q.shape)} {q.dtype}")
        B, Hq, D = q.shape
        if k_store.dim() != 4 or v_store.dim() != 4 or k_store.shape[:3] != v_store.shape[:3] or k_store.size(0) != B:
            raise _lib.KvqError("kvq: decode_step stores must be [B,Hkv,Tcap,Dq] views of one layer")
        if k_store.dtype != QDTYPE[k_kind] or v_store.dtype != QDTYPE[v_kind]:
            raise _lib.KvqError("kvq: decode_step store dtype does not match its kind")
        if k_store.size(3) != packed_dim(k_kind, D) or v_store.size(3) != packed_dim(v_kind, D):
            raise _lib.KvqError("kvq: decode_step store last dim does not match head_dim")
        if k_store.stride(3) != 1 or v_store.stride(3) != 1 or k_scales.dtype != torch.float32 or v_scales.dtype != torch.float32:
            raise _lib.KvqError("kvq: decode_step stores must be row-contiguous, scales float32")
        if k_scales.dim() != 1 or v_scales.dim() != 1 or k_scales.stride(0) != 1 or v_scales.stride(0) != 1:
            raise _lib.KvqError("kvq: decode_step scales must be contiguous [Tcap] rows")
        self.cap = min(k_store.size(2), v_store.size(2), k_scales.numel(), v_scales.numel())
        self.B, self.Hq, self.Hkv, self.D = B, Hq, k_store.size(1), D
        self.k_ptr, self.v_ptr = c_void_p(k_store.data_ptr()), c_void_p(v_store.data_ptr())
        self.ks_ptr, self.vs_ptr = c_void_p(k_scales.data_ptr()), c_void_p(v_scales.data_ptr())
        self.kst = KvqStrides(0, k_store.stride(0), k_store.stride(1), k_store.stride(2))
        self.vst = KvqStrides(0, v_store.stride(0), v_store.stride(1), v_store.stride(2))
        self.kbits, self.vbits = KIND_BITS[k_kind], KIND_BITS[v_kind]
        self.dtype_code = dtype_code(q.dtype)
        self.eps = float(eps)
        self.key = (k_store.data_ptr(), v_store.data_ptr(), q.dtype, B, Hq, D)
        self.keep = (k_store, v_store, k_scales, v_scales)


def decode_step(plan: DecodeStepPlan, q: torch.Tensor, k_new: torch.Tensor, v_new: torch.Tensor, T: int,
                out: torch.Tensor, workspace: torch.Tensor, sm_scale: float) -> None:
    """Attention over the ``T`` stored tokens + the new token, then the new token quantised into slot
    ``T`` of the stores: one layer's whole decode step in one call (kvq_decode_step). q / out
    ``[B,Hq,D]``, k_new / v_new ``[B,Hkv,D]`` (last dim contiguous). The caller bumps its token
    count afterwards."""
    if T < 0 or T >= plan.cap:
        raise _lib.KvqError(f"kvq: decode_step slot {T} is outside the store (capacity {plan.cap})")
    if (q.shape != (plan.B, plan.Hq, plan.D) or out.shape != q.shape or k_new.shape != (plan.B, plan.Hkv, plan.D)
            or v_new.shape != k_new.shape or not (q.is_cuda and k_new.is_cuda and v_new.is_cuda and out.is_cuda)
            or q.stride(2) != 1 or k_new.stride(2) != 1 or v_new.stride(2) != 1 or out.stride(2) != 1
            or k_new.dtype != q.dtype or v_new.dtype != q.dtype or out.dtype != q.dtype):
        raise _lib.KvqError("kvq: decode_step tensors do not match the plan (shape / dtype / device / contiguity)")
    dims = _lib.KvqAttnDims(plan.B, plan.Hq, plan.Hkv, T, plan.D)
    check(_launch(q.device, _lib.load().kvq_decode_step, 
        c_void_p(q.data_ptr()), q.stride(0), q.stride(1),
        c_void_p(k_new.data_ptr()), k_new.stride(0), k_new.stride(1),
        c_void_p(v_new.data_ptr()), v_new.stride(0), v_new.stride(1),
        plan.k_ptr, byref(plan.kst), plan.ks_ptr, plan.kbits, plan.v_ptr, byref(plan.vst), plan.vs_ptr, plan.vbits,
        c_void_p(out.data_ptr()), out.stride(0), out.stride(1), plan.dtype_code, float(sm_scale), plan.eps,
        c_void_p(workspace.data_ptr()), workspace.numel(), byref(dims)), "decode_step")


def decode_step_dev(plan: DecodeStepPlan, q: torch.Tensor, k_new: torch.Tensor, v_new: torch.Tensor, t_dev: torch.Tensor,
                    t_bound: int, out: torch.Tensor, workspace: torch.Tensor, sm_scale: float) -> None:
    """:func:`decode_step` with the stored-token count in device memory (kvq_decode_step_dev): ``t_dev`` is a
    one-element int32 GPU tensor, ``t_bound`` the host's upper bound on it (sizes the grid; the workspace must
    cover ``decode_attn_workspace(..., t_bound, ...)``). Every launch argument is step-independent, so the call
    can be captured into a HIP graph and replayed; the caller adds 1 to ``t_dev`` on the same stream afterwards
    and guarantees ``t_dev <= t_bound < capacity``."""
    t_bound = int(t_bound)
    if t_bound < 1 or t_bound >= plan.cap:
        raise _lib.KvqError(f"kvq: decode_step_dev bound {t_bound} is outside the store (capacity {plan.cap})")
    require_gpu(t_dev, "t_dev")
    if t_dev.dtype != torch.int32 or t_dev.numel() != 1:
        raise _lib.KvqError("kvq: decode_step_dev t_dev must be a one-element int32 tensor")
    if (q.shape != (plan.B, plan.Hq, plan.D) or out.shape != q.shape or k_new.shape != (plan.B, plan.Hkv, plan.D)
            or v_new.shape != k_new.shape or not (q.is_cuda and k_new.is_cuda and v_new.is_cuda and out.is_cuda)
            or q.stride(2) != 1 or k_new.stride(2) != 1 or v_new.stride(2) != 1 or out.stride(2) != 1
            or k_new.dtype != q.dtype or v_new.dtype != q.dtype or out.dtype != q.dtype):
        raise _lib.KvqError("kvq: decode_step_dev tensors do not match the plan (shape / dtype / device / contiguity)")
    dims = _lib.KvqAttnDims(plan.B, plan.Hq, plan.Hkv, t_bound, plan.D)
    check(_launch(q.device, _lib.load().kvq_decode_step_dev, 
        c_void_p(q.data_ptr()), q.stride(0), q.stride(1),
        c_void_p(k_new.data_ptr()), k_new.stride(0), k_new.stride(1),
        c_void_p(v_new.data_ptr()), v_new.stride(0), v_new.stride(1),
        plan.k_ptr, byref(plan.kst), plan.ks_ptr, plan.kbits, plan.v_ptr, byref(plan.vst), plan.vs_ptr, plan.vbits,
        c_void_p(out.data_ptr()), out.stride(0), out.stride(1), plan.dtype_code, float(sm_scale), plan.eps,
        c_void_p(workspace.data_ptr()), workspace.numel(), byref(dims), c_void_p(t_dev.data_ptr())), "decode_step_dev")


class DecodeLayersPlan:
    """Pointer tables of :func:`decode_step_layers`: one decode step's attention for ALL layers of a
    store behind one host call (kvq_decode_step_layers). Built once per (store allocation, query /
    output buffers); the per-step cost is one ctypes call that enqueues one launch per layer.

    q / out ``[L,B,Hq,D]``, k_new / v_new ``[L,B,Hkv,D]`` (fp16 / bf16; every layer the same strides),
    k_store / v_store ``[L,B,Hkv,Tcap,Dq]``, k_scales / v_scales ``[L,Tcap]`` fp32."""

    __slots__ = ("L", "B", "Hq", "Hkv", "D", "cap", "tabs", "q_st", "kn_st", "vn_st", "o_st", "kst", "vst", "kbits", "vbits",
                 "dtype_code", "eps", "has_new", "keep", "device")

    def __init__(self, q: torch.Tensor, k_new, v_new, out: torch.Tensor, k_store: torch.Tensor, k_scales: torch.Tensor,
                 k_kind: str, v_store: torch.Tensor, v_scales: torch.Tensor, v_kind: str, eps: float = 1e-8):
        for name, t in (("q", q), ("out", out), ("k_store", k_store), ("v_store", v_store), ("k_scales", k_scales),
                        ("v_scales", v_scales)):
            require_gpu(t, name)
        if q.dim() != 4 or out.shape != q.shape or out.dtype != q.dtype or q.dtype not in (torch.float16, torch.bfloat16):
            raise _lib.KvqError(f"kvq: decode_step_layers q / out must be matching [L,B,Hq,D] fp16 / bf16 tensors, got {tuple(q.shape)}")
        L, B, Hq, D = q.shape
        if k_store.dim() != 5 or v_store.dim() != 5 or k_store.shape[:4] != v_store.shape[:4] or k_store.shape[:2] != (L, B):
            raise _lib.KvqError("kvq: decode_step_layers stores must be [L,B,Hkv,Tcap,Dq]")
        Hkv = k_store.size(2)
        if k_store.dtype != QDTYPE[k_kind] or v_store.dtype != QDTYPE[v_kind] or k_store.size(4) != packed_dim(k_kind, D) \
                or v_store.size(4) != packed_dim(v_kind, D) or k_store.stride(4) != 1 or v_store.stride(4) != 1:
            raise _lib.KvqError("kvq: decode_step_layers store dtype / last dim does not match its kind and head_dim")
        for name, t in (("k_scales", k_scales), ("v_scales", v_scales)):
            if t.dtype != torch.float32 or t.dim() != 2 or t.size(0) != L or t.stride(1) != 1:
                raise _lib.KvqError(f"kvq: decode_step_layers {name} must be fp32 [L,Tcap] rows")
        if q.stride(3) != 1 or out.stride(3) != 1:
            raise _lib.KvqError("kvq: decode_step_layers q / out last dim must be contiguous")
        self.has_new = k_new is not None
        if (k_new is None) != (v_new is None):
            raise _lib.KvqError("kvq: decode_step_layers k_new and v_new go together")
        if self.has_new:
            for name, t in (("k_new", k_new), ("v_new", v_new)):
                require_gpu(t, name)
                if tuple(t.shape) != (L, B, Hkv, D) or t.dtype != q.dtype or t.stride(3) != 1:
                    raise _lib.KvqError(f"kvq: decode_step_layers {name} must be [L,B,Hkv,D] of the query dtype")
        self.L, self.B, self.Hq, self.Hkv, self.D = L, B, Hq, Hkv, D
        self.cap = min(k_store.size(3), v_store.size(3), k_scales.size(1), v_scales.size(1))

        def table(t):
            return _lib.ptr_array([t[i].data_ptr() for i in range(L)])

        self.tabs = {"q": table(q), "out": table(out), "k": table(k_store), "v": table(v_store), "ks": table(k_scales),
                     "vs": table(v_scales), "kn": table(k_new) if self.has_new else None,
                     "vn": table(v_new) if self.has_new else None}
        self.q_st, self.o_st = (q.stride(1), q.stride(2)), (out.stride(1), out.stride(2))
        self.kn_st = (k_new.stride(1), k_new.stride(2)) if self.has_new else (0, 0)
        self.vn_st = (v_new.stride(1), v_new.stride(2)) if self.has_new else (0, 0)
        self.kst = KvqStrides(0, k_store.stride(1), k_store.stride(2), k_store.stride(3))
        self.vst = KvqStrides(0, v_store.stride(1), v_store.stride(2), v_store.stride(3))
        self.kbits, self.vbits = KIND_BITS[k_kind], KIND_BITS[v_kind]
        self.dtype_code = dtype_code(q.dtype)
        self.eps = float(eps)
        self.device = q.device
        self.keep = (q, k_new, v_new, out, k_store, k_scales, v_store, v_scales)


def decode_step_layers(plan: DecodeLayersPlan, T: int, workspace: torch.Tensor, sm_scale: float, append: bool = False) -> None:
    """Every layer's decode attention over its first ``T`` stored tokens (+ the plan's new token) in ONE
    host call; ``append`` also quantises the new token into slot ``T`` of every layer's store
    (kvq_decode_step per layer) — the caller bumps its token count afterwards."""
    T = int(T)
    if T < 0 or T > plan.cap or (append and (T >= plan.cap or not plan.has_new)):
        raise _lib.KvqError(f"kvq: decode_step_layers T={T} does not fit the store (capacity {plan.cap}) or append without a new token")
    require_gpu(workspace, "workspace")
    if workspace.dtype != torch.float32 or not workspace.is_contiguous():
        raise _lib.KvqError("kvq: decode_step_layers workspace must be contiguous float32")
    t = plan.tabs
    dims = _lib.KvqAttnDims(plan.B, plan.Hq, plan.Hkv, T, plan.D)
    check(_launch(plan.device, _lib.load().kvq_decode_step_layers, 
        plan.L, 1 if append else 0, t["q"], plan.q_st[0], plan.q_st[1], t["kn"], plan.kn_st[0], plan.kn_st[1],
        t["vn"], plan.vn_st[0], plan.vn_st[1], t["k"], byref(plan.kst), t["ks"], plan.kbits, t["v"], byref(plan.vst), t["vs"],
        plan.vbits, t["out"], plan.o_st[0], plan.o_st[1], plan.dtype_code, float(sm_scale), plan.eps,
        c_void_p(workspace.data_ptr()), workspace.numel(), byref(dims)), "decode_step_layers")
