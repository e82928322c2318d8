"""CPU oracle (numpy) for the KV-cache quantize / dequantize / eviction hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and there only as the checker.  The product path
(``efficient-llm-inference_amd/``) never imports this module and fails loudly when the
HIP library is missing.

Parity status: PINNED.  Every function below is checked in ``tests/test_oracle_golden.py``
against golden vectors captured from the reference itself, imported in the build
container (generator: ``tests/golden/make_golden.py``; fixtures: ``tests/golden/*.npz``).

Each function restates one reference function; citations are ``path:line`` relative to
the reference checkout.  Arithmetic is IEEE fp32 exactly as the reference's torch ops
perform it (``.float()``, ``abs().max()``, true division, ``round`` = half-to-even,
``clamp``, ``.to(dtype)`` = round-to-nearest-even).

numpy has no bfloat16: bf16 tensors travel as ``uint16`` bit patterns together with
``dtype="bf16"`` (helpers ``bf16_bits_to_f32`` / ``f32_to_bf16_bits``).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

# --------------------------------------------------------------------------- dtypes


def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    """uint16 bf16 bit patterns -> float32 (exact)."""
    return (bits.astype(np.uint32) << np.uint32(16)).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit patterns, round-to-nearest-even (what ``.to(torch.bfloat16)`` does).

    NaN inputs are not on the path and are not handled specially.
    """
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)
    return r.astype(np.uint16)


def _widen(x: np.ndarray, dtype: str | None) -> np.ndarray:
    """``x.float()`` (ops.py:26, ops.py:47)."""
    if dtype == "bf16":
        return bf16_bits_to_f32(x)
    return x.astype(np.float32)


def _round_to_storage(v: np.ndarray, like: np.ndarray, dtype: str | None) -> np.ndarray:
    """``t.to(x.dtype)`` (ops.py:30, ops.py:65): RN-even to the storage dtype."""
    if dtype == "bf16":
        return f32_to_bf16_bits(np.asarray(v, dtype=np.float32))
    return np.asarray(v, dtype=np.float32).astype(like.dtype)


def stored_scale_as_f32(scale: np.ndarray, dtype: str | None = None) -> np.ndarray:
    """``scale.float()`` of the *stored* scale (ops.py:87, :90, :117, :133)."""
    if dtype == "bf16":
        return bf16_bits_to_f32(np.asarray(scale))
    return np.asarray(scale).astype(np.float32)


# ----------------------------------------------------------- a1 / a2: per-tensor quantise


def _scale_f32(max_abs: np.ndarray, qmax: float, eps: float) -> np.ndarray:
    """``(max_abs / qmax).clamp(min=eps)`` in fp32 (ops.py:28, ops.py:49)."""
    return np.maximum(np.asarray(max_abs, dtype=F32) / F32(qmax), F32(eps)).astype(F32)


def quantize_int8_per_tensor(x: np.ndarray, eps: float = 1e-8, dtype: str | None = None):
    """Reference ``quantize_int8_per_tensor`` (src/quantization/ops.py:10-30).

    Returns ``(q int8 same shape, scale 0-dim in x's dtype)``.  The fp32 scale is used
    for the quantisation, the returned scale is rounded to the input dtype.
    """
    x32 = _widen(x, dtype)
    max_abs = np.abs(x32).max() if x32.size else F32(0)
    s32 = _scale_f32(max_abs, 127.0, eps)
    q = np.clip(np.rint(x32 / s32), -127, 127).astype(np.int8)
    return q, _round_to_storage(s32, x, dtype)


def pack_int4(q: np.ndarray) -> np.ndarray:
    """Nibble packing of ops.py:52-63: pad odd last dim with one zero (ops.py:54-56),
    ``u = q + 8`` (ops.py:59), even index -> HIGH nibble, odd index -> low (ops.py:61-63)."""
    if q.shape[-1] % 2 == 1:
        q = np.concatenate([q, np.zeros_like(q[..., :1])], axis=-1)
    u = (q.astype(np.int16) + 8).astype(np.uint8)
    return ((u[..., 0::2] << np.uint8(4)) | u[..., 1::2]).astype(np.uint8)


def unpack_int4(packed: np.ndarray, orig_last_dim: int) -> np.ndarray:
    """Inverse of :func:`pack_int4` (ops.py:122-132; extensions.py:60-62): int8 in [-8, 7]."""
    hi = (packed >> np.uint8(4)) & np.uint8(0x0F)
    lo = packed & np.uint8(0x0F)
    u = np.empty(packed.shape[:-1] + (packed.shape[-1] * 2,), dtype=np.uint8)
    u[..., 0::2] = hi
    u[..., 1::2] = lo
    return (u.astype(np.int16) - 8).astype(np.int8)[..., :orig_last_dim]


def quantize_int4_per_tensor_packed(x: np.ndarray, eps: float = 1e-8, dtype: str | None = None):
    """Reference ``quantize_int4_per_tensor_packed`` (src/quantization/ops.py:33-65).

    Returns ``(packed uint8 [..., ceil(D/2)], scale in x's dtype, orig_last_dim)``.
    """
    x32 = _widen(x, dtype)
    max_abs = np.abs(x32).max() if x32.size else F32(0)
    s32 = _scale_f32(max_abs, 7.0, eps)
    q = np.clip(np.rint(x32 / s32), -8, 7).astype(np.int8)
    return pack_int4(q), _round_to_storage(s32, x, dtype), int(x.shape[-1])


# --------------------------------------------------------- a3 / a4: per-tensor dequantise


def _to_out(v32: np.ndarray, out_dtype: str) -> np.ndarray:
    if out_dtype == "f16":
        return v32.astype(np.float16)
    if out_dtype == "bf16":
        return f32_to_bf16_bits(v32)
    if out_dtype == "f32":
        return v32.astype(np.float32)
    raise ValueError(out_dtype)


def dequantize_int8_per_tensor(q: np.ndarray, scale_f32, out_dtype: str = "f16") -> np.ndarray:
    """Reference ``dequantize_int8_per_tensor`` (ops.py:68-90) and the CUDA kernel it may call
    (src/cuda/extensions.py:37-48): ``out = RN_out(float(q) * float(scale))``.

    ``scale_f32`` is the stored scale widened to fp32 (:func:`stored_scale_as_f32`).
    """
    v = q.astype(np.float32) * F32(scale_f32)
    return _to_out(v.astype(np.float32), out_dtype)


def dequantize_int4_per_tensor_packed(
    packed: np.ndarray, scale_f32, orig_last_dim: int, out_dtype: str = "f16"
) -> np.ndarray:
    """Reference ``dequantize_int4_per_tensor_packed`` (ops.py:93-133) and its CUDA kernel
    (extensions.py:50-68): high nibble first, ``(nibble - 8) * scale``, pad column dropped."""
    q = unpack_int4(packed, orig_last_dim)
    v = q.astype(np.float32) * F32(scale_f32)
    return _to_out(v.astype(np.float32), out_dtype)


def dequant_int4_packed_kernel_full(packed: np.ndarray, scale_f32, orig_last_dim: int) -> np.ndarray:
    """The raw CUDA entry ``dequant_int4_packed_to_fp16`` BEFORE the python slice
    (extensions.py:88-114): last dim = 2*packed_last, pad column written as 0 (:65-66)."""
    total = packed.shape[-1] * 2
    q = unpack_int4(packed, total)
    v = (q.astype(np.float32) * F32(scale_f32)).astype(np.float16)
    v[..., orig_last_dim:] = np.float16(0)
    return v


# ------------------------------------------- a5 / a6: token-wise cache (vectorised restatement)
# QuantizedLayerKV.append (ops.py:174-210) quantises one slice [B,H,1,D] per token per K|V
# with ONE scale over the whole slice.  Vectorised: reduce abs-max over axes (B,H,D) for
# every (g, t) of a [G,B,H,T,D] tensor.  G is any leading batch of independent groups
# (layer x K|V); it never shares a scale.

QMAX = {"int8": 127.0, "int4": 7.0}
QMIN = {"int8": -127, "int4": -8}


def quantize_tokens(x: np.ndarray, kind: str, eps: float = 1e-8, dtype: str | None = None):
    """[G,B,H,T,D] -> (q, scales_stored[G,T], scales_f32[G,T]).

    kind="int8": q int8 [G,B,H,T,D]; kind="int4": q uint8 [G,B,H,T,ceil(D/2)].
    Equivalent to calling a1/a2 on every ``x[g,:,:,t:t+1,:]`` (ops.py:339-342).
    """
    assert x.ndim == 5
    x32 = _widen(x, dtype)
    max_abs = np.abs(x32).max(axis=(1, 2, 4)) if x32.size else np.zeros((x.shape[0], x.shape[3]), F32)
    s32 = _scale_f32(max_abs, QMAX[kind], eps)  # [G,T]
    q = np.clip(np.rint(x32 / s32[:, None, None, :, None]), QMIN[kind], QMAX[kind]).astype(np.int8)
    if kind == "int4":
        q = pack_int4(q)
    stored = _round_to_storage(s32, x, dtype)
    return q, stored, stored_scale_as_f32(stored, dtype)


def absmax_tokens(x: np.ndarray, dtype: str | None = None) -> np.ndarray:
    """[G,B,H,T,D] -> [G,T] fp32 ``max |x|`` over (B,H,D): the ``x32.abs().max()`` of ops.py:27 / :48 taken
    over the rows at hand. For a batch split over ranks, the element-wise MAX of the ranks' tables IS the
    abs-max of the whole slice (max is associative and exact), which is what the sharded path exchanges."""
    assert x.ndim == 5
    x32 = _widen(x, dtype)
    return np.abs(x32).max(axis=(1, 2, 4)).astype(F32) if x32.size else np.zeros((x.shape[0], x.shape[3]), F32)


def quantize_tokens_with_absmax(x: np.ndarray, max_abs: np.ndarray, kind: str, eps: float = 1e-8, dtype: str | None = None):
    """:func:`quantize_tokens` with the abs-max table GIVEN (ops.py:28-30 / :49-65 after the max): the rows of a
    sharded batch quantised with the whole batch's scale. Returns (q, scales_stored, scales_f32)."""
    x32 = _widen(x, dtype)
    s32 = _scale_f32(np.asarray(max_abs, dtype=F32), QMAX[kind], eps)
    q = np.clip(np.rint(x32 / s32[:, None, None, :, None]), QMIN[kind], QMAX[kind]).astype(np.int8)
    if kind == "int4":
        q = pack_int4(q)
    stored = _round_to_storage(s32, x, dtype)
    return q, stored, stored_scale_as_f32(stored, dtype)


def dequantize_tokens(q: np.ndarray, scales_f32: np.ndarray, kind: str, D: int, out_dtype: str = "f16"):
    """Inverse of :func:`quantize_tokens`: what ``get_kv`` (ops.py:213-269) returns after the
    T-way ``torch.cat(dim=2)``.  q [G,B,H,T,Dq], scales_f32 [G,T] -> [G,B,H,T,D]."""
    if kind == "int4":
        q = unpack_int4(q, D)
    v = q.astype(np.float32) * scales_f32.astype(np.float32)[:, None, None, :, None]
    return _to_out(v.astype(np.float32), out_dtype)


def estimated_bytes(mode: str, L: int, B: int, H: int, T: int, D: int, scale_itemsize: int) -> int:
    """``QuantizedKVCache.estimated_bytes`` (ops.py:271-290, :357-363): stores + scales,
    scale itemsize follows the input dtype (ops.py:30,65)."""
    per_tok = {"int8": B * H * D, "int4": B * H * ((D + 1) // 2)}
    k_kind, v_kind = {"int8": ("int8", "int8"), "int4": ("int4", "int4"), "mixed": ("int8", "int4")}[mode]
    return L * T * (per_tok[k_kind] + per_tok[v_kind] + 2 * scale_itemsize)


# ------------------------------------------------------------------------ a7: sliding window


def trim_kv_sliding_window(x: np.ndarray, window_size: int) -> np.ndarray:
    """Reference ``trim_kv_sliding_window`` (src/cache/implementations.py:124-140) on one
    [..., T, D] tensor: last ``window_size`` tokens if T > window, else unchanged."""
    T = x.shape[-2]
    if T > window_size and window_size != 0:  # window_size == 0: the reference's `-0:` slice is the whole tensor (:137-139)
        return x[..., T - window_size :, :]
    return x


# ------------------------------------------------------------------ a8: chunk-summary pooling


def chunk_summary_len(T: int, chunk_size: int, keep_last: int) -> int:
    """Output length of ``chunk_summarize_kv`` (implementations.py:313-345)."""
    keep = min(keep_last, T)
    old = T - keep
    if old <= 0:
        return T
    return (old + chunk_size - 1) // chunk_size + keep


def chunk_summarize_kv(x: np.ndarray, chunk_size: int, keep_last: int, dtype: str | None = None) -> np.ndarray:
    """Reference ``chunk_summarize_kv`` (implementations.py:295-346) on one [..., T, D] tensor.

    Older tokens are zero-padded to a multiple of ``chunk_size`` (:326-333) and mean-pooled
    (:338-339); the divisor is ``chunk_size`` even for the padded last chunk.  Accumulation is
    fp32, SEQUENTIAL over the chunk's tokens in increasing t (the order the HIP kernel uses),
    then one true division by ``chunk_size`` and one RN to the storage dtype.  torch's CPU
    ``mean`` uses its own (cascade) summation order, so this restatement is pinned to the
    reference's golden vectors within 1 storage-dtype ulp, not bit-exactly (tests say so).
    """
    T, D = x.shape[-2], x.shape[-1]
    keep = min(keep_last, T)
    old = T - keep
    if old <= 0:
        return x
    n_chunks = (old + chunk_size - 1) // chunk_size
    x32 = _widen(x, dtype)
    acc = np.zeros(x.shape[:-2] + (n_chunks, D), dtype=F32)
    for j in range(chunk_size):
        idx = np.arange(n_chunks) * chunk_size + j
        valid = idx < old
        if not valid.any():
            break
        acc[..., valid, :] = (acc[..., valid, :] + x32[..., idx[valid], :]).astype(F32)
    pooled = (acc / F32(chunk_size)).astype(F32)
    pooled_st = _round_to_storage(pooled, x, dtype)
    return np.concatenate([pooled_st, x[..., old:, :]], axis=-2)


# ------------------------------------------------ N3: index-select eviction family + paged layout
# All four policies keep [optional prefix] + [selected older tokens] + [dense tail]; they differ
# only in which older tokens are selected. Restated as index lists (host integers).


def _tail_start(T: int, prefix_len: int, window_size: int) -> int:
    return max(prefix_len, T - window_size)


def keep_indices_prefix_window(T: int, prefix_len: int, window_size: int):
    """``trim_kv_prefix_window`` (src/cache/implementations.py:143-154): first prefix_len + last
    window_size tokens; None = unchanged (T <= prefix_len + window_size)."""
    if T <= prefix_len + window_size:
        return None
    # window_size == 0: `k[:, :, -0:, :]` is the WHOLE tensor (:151-152): prefix followed by every token
    return list(range(prefix_len)) + (list(range(T)) if window_size == 0 else list(range(T - window_size, T)))


def keep_indices_strided(T: int, window_size: int, stride: int, prefix_len: int = 0):
    """``trim_kv_strided`` (implementations.py:157-190)."""
    assert stride >= 1
    if T <= prefix_len + window_size:
        return None
    ts = _tail_start(T, prefix_len, window_size)
    return list(range(prefix_len)) + list(range(prefix_len, ts, stride)) + list(range(ts, T))


def keep_indices_block_old(T: int, window_size: int, block_size: int = 64, keep_per_block: int = 8, prefix_len: int = 0):
    """``trim_kv_block_old`` (implementations.py:193-245): last keep_per_block tokens of every
    block_size-block of the older region (the ragged last block included)."""
    assert block_size >= 1 and 1 <= keep_per_block <= block_size
    if T <= prefix_len + window_size:
        return None
    ts = _tail_start(T, prefix_len, window_size)
    old = []
    start = prefix_len
    while start < ts:
        end = min(start + block_size, ts)
        old += list(range(max(start, end - keep_per_block), end))
        start = end
    return list(range(prefix_len)) + old + list(range(ts, T))


def keep_indices_budget_old(T: int, window_size: int, old_budget: int = 64, prefix_len: int = 0):
    """``trim_kv_budget_old`` (implementations.py:248-292): old_budget tokens sampled with
    ``torch.linspace(prefix_len, tail_start-1, steps=old_budget).long()`` (fp32 linspace, truncation)
    then ``unique_consecutive`` (:279-282). numpy restatement of torch's symmetric fp32 linspace."""
    assert old_budget >= 0
    if T <= prefix_len + window_size:
        return None
    ts = _tail_start(T, prefix_len, window_size)
    old_len = ts - prefix_len
    old = []
    if old_len > 0 and old_budget > 0:
        if old_len <= old_budget:
            old = list(range(prefix_len, ts))
        else:
            start, end, steps = np.float32(prefix_len), np.float32(ts - 1), old_budget
            if steps == 1:
                vals = np.array([start], dtype=np.float32)
            else:
                step = np.float32((end - start) / np.float32(steps - 1))
                i = np.arange(steps)
                half = steps // 2
                lo = (start + step * i.astype(np.float32)).astype(np.float32)
                hi = (end - step * (steps - 1 - i).astype(np.float32)).astype(np.float32)
                vals = np.where(i < half, lo, hi).astype(np.float32)
            idx = vals.astype(np.int64)  # .long(): truncation toward zero (values are >= 0)
            keep = np.ones(len(idx), dtype=bool)
            keep[1:] = idx[1:] != idx[:-1]  # unique_consecutive
            old = idx[keep].tolist()
    return list(range(prefix_len)) + old + list(range(ts, T))


def gather_tokens(x: np.ndarray, idx) -> np.ndarray:
    """index_select along the token axis of [..., T, D]; idx None = unchanged."""
    if idx is None:
        return x
    return x[..., np.asarray(idx, dtype=np.int64), :]


# ----------------------------------------------------------------------------------------------
# N1 (second form): one decode step of attention over the quantised cache


def decode_attention(q, k_q, k_scales_f32, k_kind, v_q, v_scales_f32, v_kind, D, sm_scale,
                     k_new=None, v_new=None, kv_dtype: str = "f16") -> np.ndarray:
    """What the reference computes for ONE layer and ONE new query token, in float64.

    ``QuantizedKVCache.to_past_key_values`` (ops.py:345-355) dequantises every stored token to the
    compute dtype (``get_kv`` :213-269); the model then cats the new token's exact k/v and runs
    ``softmax(q K^T * sm_scale) V`` (HF attention under benchmarker.py:470-471).

    q [B,Hq,D]; k_q / v_q [B,Hkv,T,Dq] (int8 or packed uint8); *_scales_f32 [T]; k_new / v_new
    [B,Hkv,D] or None. Grouped-query: query head h reads kv head h // (Hq // Hkv).
    Returns float64 [B,Hq,D] (callers compare at the compute dtype's tolerance).
    """
    q = np.asarray(q, dtype=np.float64)
    B, Hq, _ = q.shape
    kd = dequantize_tokens(k_q[None], np.asarray(k_scales_f32, F32)[None], k_kind, D, kv_dtype)[0]
    vd = dequantize_tokens(v_q[None], np.asarray(v_scales_f32, F32)[None], v_kind, D, kv_dtype)[0]
    kd = (bf16_bits_to_f32(kd) if kv_dtype == "bf16" else kd).astype(np.float64)
    vd = (bf16_bits_to_f32(vd) if kv_dtype == "bf16" else vd).astype(np.float64)
    if k_new is not None:
        kd = np.concatenate([kd, np.asarray(k_new, np.float64)[:, :, None, :]], axis=2)
        vd = np.concatenate([vd, np.asarray(v_new, np.float64)[:, :, None, :]], axis=2)
    Hkv = kd.shape[1]
    rep = Hq // Hkv
    out = np.empty((B, Hq, D), np.float64)
    for b in range(B):
        for h in range(Hq):
            s = kd[b, h // rep] @ q[b, h] * float(sm_scale)  # [T(+1)]
            p = np.exp(s - s.max())
            out[b, h] = (p / p.sum()) @ vd[b, h // rep]
    return out
