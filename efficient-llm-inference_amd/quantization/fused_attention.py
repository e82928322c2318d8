"""Decode with attention computed straight from the quantised store (scope row N1, second form).

The reference dequantises the whole cache to fp16 before every forward
(``QuantizedKVCache.to_past_key_values``, reference src/quantization/ops.py:345-355, called from
src/benchmarking/benchmarker.py:470-471) and the model's attention then reads that fp16 copy. Here
the model's attention function IS the HIP kernel ``kvq_decode_attn``: it reads the INT8 / packed
INT4 rows and their per-token scales, adds the new token's exact K/V as one more softmax term
(what the reference's ``cat`` does), and no fp16 copy of the cache exists at any time — the
resident KV really is ``estimated_bytes()``.

Plumbing (transformers >= 4.54 / 5.x):

  * :class:`FusedQuantizedCache` owns a :class:`QuantizedKVCache` and hands the model a
    ``DynamicCache`` whose layers quantise the prompt's K/V into the store (2 launches per layer) and
    pass the new tokens' exact K/V on to the attention function; a decode token is quantised by the
    same host call that attends (``kvq_decode_step``: one ctypes call and, for grouped-query shapes,
    one launch per layer and step);
  * ``kvq_fused`` is registered with transformers' ``AttentionInterface``; inside
    :func:`fused_attention` the model's ``_attn_implementation`` points at it. A single-token
    query runs ``kernels.decode_attn`` on the layer's store; a prompt (empty cache) runs exact
    causal SDPA over the prompt's own K/V, as the reference's un-quantised prompt forward does.

Numerics: fp32 accumulation over exact integer x fp16 products; the rounding of each dequantised
value to fp16 that the reference's tuple path performs is skipped, so logits agree within fp16
tolerance, not bit-for-bit (tests/test_gpu_attn.py, tests/test_gpu_benchmarker.py).
Limits (fail loudly): fp16 / bf16 models, head_dim in {32, 64, 128, 256}, at most 8 query heads
per kv head (16 at head_dim 64 / 128), no padding mask during decode, no chunked prefill into a non-empty cache,
no sliding-window / soft-capped / attention-sink layers (``_reject_unsupported_variants``).
"""
from __future__ import annotations

import contextlib
from typing import Optional

import torch

from .. import kernels
from .ops import QuantizedKVCache

try:
    from transformers import AttentionInterface
    from transformers.cache_utils import DynamicCache, DynamicLayer
    from transformers.masking_utils import AttentionMaskInterface, sdpa_mask
except Exception:  # pragma: no cover - older transformers: the staged / tuple paths are used instead
    AttentionInterface = None
    DynamicCache = DynamicLayer = None

ATTN_NAME = "kvq_fused"
_ACTIVE: Optional["FusedQuantizedCache"] = None


def available() -> bool:
    return AttentionInterface is not None and DynamicLayer is not None


if DynamicLayer is not None:

    class _FusedLayer(DynamicLayer):
        """One layer of the HF cache, backed by group ``index`` of the K and V stores."""

        def __init__(self, owner: "FusedQuantizedCache", index: int):
            super().__init__()
            self._owner = owner
            self._index = index
            self.is_initialized = True
            self.past = 0  # tokens in the store before the forward in flight
            self.pending = False  # the new token still has to be quantised (done by the attention call)
            self.plan = None  # kernels.DecodeStepPlan of this layer
            self.keys = self.values = None  # nothing dequantised is ever kept

        def update(self, key_states: torch.Tensor, value_states: torch.Tensor, *args, **kwargs):
            qc, i = self._owner.qcache, self._index
            self.past = qc._k.lens[i]
            self.dtype, self.device = key_states.dtype, key_states.device
            if key_states.shape[-2] == 1 and self.past > 0:
                # decode: the attention function quantises this token into slot `past` in the same
                # host call that attends (kvq_decode_step); only the capacity is settled here
                if self._owner.t_dev is None:  # graph mode: capacity was reserved up front, nothing may reallocate
                    qc._k.reserve(self.past + 1)
                    qc._v.reserve(self.past + 1)
                self.pending = True
                return key_states, value_states
            # prompt: a dense copy takes the single-pass quantise path; one launch per store
            qc._k.append([key_states.contiguous()], g0=i)
            qc._v.append([value_states.contiguous()], g0=i)
            self.pending = False
            return key_states, value_states  # the new tokens' exact K/V go on to the attention function

        def get_seq_length(self) -> int:
            return self._owner.qcache._k.lens[self._index]


class FusedQuantizedCache:
    """A :class:`QuantizedKVCache` the model appends to and attends over directly."""

    def __init__(self, n_layers: int, mode: str = "int8", device: str = "cuda",
                 compute_dtype: torch.dtype = torch.float16, reserve: int = 0):
        if not available():
            raise RuntimeError("kvq: this transformers version has no AttentionInterface / DynamicLayer")
        self.qcache = QuantizedKVCache(n_layers=n_layers, mode=mode, device=device, compute_dtype=compute_dtype)
        if reserve:
            self.qcache.reserve(reserve)
        self.cache = DynamicCache()
        self.cache.layers = [_FusedLayer(self, i) for i in range(n_layers)]
        self._ws: Optional[torch.Tensor] = None
        self._ws_key = None
        self._ws_T = 0
        # transformers materialises the (all-true) causal mask of a single-token query while a stream is being
        # captured instead of skipping it; a driver that guarantees un-padded prompts (the graphed decode of
        # KVCacheBenchmarker) sets this so that the mask is ignored rather than refused
        self.ignore_decode_mask = False
        # graph mode (KVCacheBenchmarker.graph_decode): the stored-token count lives in this one-element int32
        # GPU tensor, every decode step runs kvq_decode_step_dev with step-independent launch arguments, and the
        # DRIVER advances the host-side lengths (once per replayed step) — see _fused_attention_forward
        self.t_dev: Optional[torch.Tensor] = None
        self.t_bound = 0

    def workspace(self, B: int, Hq: int, Hkv: int, T: int, D: int, device) -> torch.Tensor:
        """Scratch for the split partials, sized once for the reserved capacity (decode never reallocates
        and asks the library for the size only when the shape or the capacity changes)."""
        key = (B, Hq, Hkv, D, self.qcache._k.cap)
        if self._ws is None or self._ws_key != key or T > self._ws_T:
            cap = max(self.qcache._k.cap, T, 1)
            # the per-T size is not monotone in T: ask for the maximum over every T <= cap
            self._ws = torch.empty(kernels.decode_attn_workspace_cap(B, Hq, Hkv, cap, D), dtype=torch.float32, device=device)
            self._ws_key, self._ws_T = key, cap
        return self._ws

    def estimated_bytes(self) -> int:
        return self.qcache.estimated_bytes()


# what transformers passes for attention variants this kernel does not implement: plain softmax over
# EVERY stored token is all it computes, so any of these must stop the run instead of being ignored
_UNSUPPORTED_KWARGS = ("sliding_window", "softcap", "s_aux", "sinks")
_UNSUPPORTED_MODULE_ATTRS = ("sliding_window", "sinks", "attn_logit_softcapping")


def _reject_unsupported_variants(module, kwargs) -> None:
    """Sliding-window (Mistral / Qwen2 / Gemma), logit soft-capping (Gemma 2) and attention sinks reach an
    attention-interface function as keyword arguments or module attributes. Raise on any of them."""
    for name in _UNSUPPORTED_KWARGS:
        if kwargs.get(name) is not None:
            raise RuntimeError(f"kvq: fused attention over the quantised store does not implement '{name}' "
                               f"(got {kwargs[name]!r}); use the staged path (fused_attention=False) for this model")
    for name in _UNSUPPORTED_MODULE_ATTRS:
        if getattr(module, name, None) is not None:
            raise RuntimeError(f"kvq: fused attention over the quantised store does not implement the attention "
                               f"variant this layer is configured with ({type(module).__name__}.{name} = "
                               f"{getattr(module, name)!r}); use the staged path (fused_attention=False)")


def _fused_attention_forward(module, query, key, value, attention_mask=None, dropout: float = 0.0,
                             scaling: Optional[float] = None, **kwargs):
    """transformers attention-interface function: ``(attn_output [B, n, Hq, D], None)``."""
    owner = _ACTIVE
    if owner is None:
        raise RuntimeError("kvq: the 'kvq_fused' attention runs only inside fused_attention(model, cache)")
    if dropout:
        raise RuntimeError("kvq: fused decode attention is inference-only (dropout must be 0)")
    _reject_unsupported_variants(module, kwargs)
    B, Hq, n, D = query.shape
    Hkv = key.shape[1]
    scale = float(scaling) if scaling is not None else D ** -0.5
    layer = owner.cache.layers[module.layer_idx]
    if n > 1 or layer.past == 0:
        # the prompt forward (or the very first token): exact attention over the tokens' own K/V,
        # like the reference's un-quantised prompt pass (benchmarker.py:445-449)
        if layer.past != 0:
            raise RuntimeError("kvq: multi-token forward into a non-empty quantised cache is not supported")
        if attention_mask is not None:
            raise RuntimeError("kvq: fused attention does not take a padding mask")
        out = torch.nn.functional.scaled_dot_product_attention(query, key, value, is_causal=n > 1, scale=scale,
                                                               enable_gqa=Hkv != Hq)
        return out.transpose(1, 2).contiguous(), None
    if attention_mask is not None and not owner.ignore_decode_mask:
        raise RuntimeError("kvq: fused decode attention does not take a padding mask")
    qc, i = owner.qcache, module.layer_idx
    T = layer.past
    out = torch.empty(B, 1, Hq, D, dtype=query.dtype, device=query.device)
    q3 = query[:, :, 0]
    plan = layer.plan
    if plan is None or plan.key != (qc._k.q[i].data_ptr(), qc._v.q[i].data_ptr(), query.dtype, B, Hq, D):
        plan = layer.plan = kernels.DecodeStepPlan(q3, qc._k.q[i], qc._k.scales[i], qc._k.kind, qc._v.q[i],
                                                   qc._v.scales[i], qc._v.kind, qc._k.eps)
    if owner.t_dev is not None:
        # device-side T: the same launches serve every later step (captured once, replayed per token); the
        # host-side lengths are the driver's to advance
        kernels.decode_step_dev(plan, q3, key[:, :, 0], value[:, :, 0], owner.t_dev, owner.t_bound, out[:, 0],
                                owner.workspace(B, Hq, Hkv, owner.t_bound, D, query.device), scale)
        layer.pending = False
        return out, None
    kernels.decode_step(plan, q3, key[:, :, 0], value[:, :, 0], T, out[:, 0],
                        owner.workspace(B, Hq, Hkv, T, D, query.device), scale)
    qc._k.lens[i] = qc._v.lens[i] = T + 1  # the step appended the token
    layer.pending = False
    return out, None


@contextlib.contextmanager
def fused_attention(model, cache: FusedQuantizedCache):
    """Route ``model``'s attention through the quantised store of ``cache`` for the duration."""
    global _ACTIVE
    if not available():
        raise RuntimeError("kvq: this transformers version has no AttentionInterface / DynamicLayer")
    AttentionInterface.register(ATTN_NAME, _fused_attention_forward)
    AttentionMaskInterface.register(ATTN_NAME, sdpa_mask)
    before = model.config._attn_implementation
    prev_active = _ACTIVE
    model.set_attn_implementation(ATTN_NAME)
    _ACTIVE = cache
    try:
        yield cache.cache
    finally:
        _ACTIVE = prev_active
        model.set_attn_implementation(before)


__all__ = ["FusedQuantizedCache", "fused_attention", "available", "ATTN_NAME"]
