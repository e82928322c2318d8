"""KVCacheBenchmarker — the caller of the hot path, with the reference's API.

Mirrors reference ``src/benchmarking/benchmarker.py``: same class name, constructor, method names,
argument names/defaults, return tuples, the 12 method names accepted by ``benchmark_method`` and
the 14-key result dict (reference benchmarker.py:643-832). The greedy decode loops keep the
reference's quirks on purpose (they are observable): the quantised / sliding / chunked loops do not
truncate the prompt, do not clamp token ids and never stop at EOS (:438, :172, :592), the
full-cache loop truncates to 1024 tokens and clamps (:116-118, :139), ``compute_dtype`` follows the
device STRING (:452), and ``benchmark_method`` ignores its ``mode`` argument for ``quant_*``
(:719-735).

What is different underneath:
  * the cache-policy calls go to the HIP kernels (quantization/ops.py, cache/implementations.py);
  * prefill quantisation is 2 launches instead of L*T Python iterations (ops.py:339-342), each
    decode step dequantises with 2 launches instead of 2*L*T (+2*L*T host syncs, ops.py:87,117);
  * a shim bridges transformers >= 5 (``DynamicCache(ddp_cache_data=...)``, ``cache.layers[i]``)
    and 4.x (``from_legacy_cache`` / ``to_legacy_cache``, which the reference calls at 32 sites).

All 12 method names of the reference are implemented; the index-select policies and the paged
layout share one row-gather / block-stitch kernel each.
"""
from __future__ import annotations

import math
import time
from typing import Callable, Tuple

import torch

from ..cache import (
    PagedKVCache,
    chunk_summarize_kv,
    trim_kv_block_old,
    trim_kv_budget_old,
    trim_kv_prefix_window,
    trim_kv_sliding_window,
    trim_kv_strided,
)
from ..core.memory import get_cpu_mem_mb, get_gpu_peak_mb, mb, reset_gpu_peak
from ..quantization import QuantizedKVCache, fused_attention, hf_cache

try:  # transformers is only needed to rebuild a Cache object for the model
    from transformers import DynamicCache
except Exception:  # pragma: no cover
    DynamicCache = None

VALID_METHODS = [
    "no_cache", "full_cache", "sliding_window", "quant_int8", "quant_int4", "quant_mixed",
    "paged_attention", "chunked_cache", "prefix_window", "strided_cache", "block_cache", "budget_cache",
]


# ----------------------------------------------------------------------------- cache-format shim


def to_legacy_tuple(past) -> tuple:
    """Any HF cache object -> ``tuple_L[(k, v)]`` of ``[B,H,T,D]`` tensors
    (reference: ``out.past_key_values.to_legacy_cache()``, e.g. benchmarker.py:445-449)."""
    if isinstance(past, (tuple, list)):
        return tuple((kv[0], kv[1]) for kv in past)
    if hasattr(past, "to_legacy_cache"):  # transformers 4.x
        return tuple((kv[0], kv[1]) for kv in past.to_legacy_cache())
    if hasattr(past, "layers"):  # transformers >= 5
        return tuple((layer.keys, layer.values) for layer in past.layers)
    raise TypeError(f"kvq: cannot read KV out of {type(past).__name__}")


def from_legacy_tuple(past_tuple: tuple):
    """``tuple_L[(k, v)]`` -> the Cache object the installed transformers expects
    (reference: ``DynamicCache.from_legacy_cache``, e.g. benchmarker.py:471)."""
    if DynamicCache is None:
        return past_tuple
    if hasattr(DynamicCache, "from_legacy_cache"):
        return DynamicCache.from_legacy_cache(past_tuple)
    return DynamicCache(ddp_cache_data=past_tuple)


class KVCacheBenchmarker:
    """Greedy-decode benchmarker over KV-cache policies (reference benchmarker.py:23-60).

    Args:
        model: HuggingFace-style causal LM (``model(input_ids=..., use_cache=..., past_key_values=...)``
            returning ``.logits`` and ``.past_key_values``)
        tokenizer: callable ``tokenizer(prompt, return_tensors="pt", ...)`` with ``.input_ids``,
            ``decode`` and ``eos_token_id``
        device: "cuda" (MI355X through PyTorch-ROCm) or "cpu" (plumbing only: no_cache / full_cache)
    """

    def __init__(self, model, tokenizer, device: str = "cuda"):
        self.model = model
        self.tokenizer = tokenizer
        self.device = device
        # quantised decode: let the model append into the cache's staging buffers in place
        # (quantization/hf_cache.py) instead of rebuilding a DynamicCache from a tuple every step.
        # Same tokens either way; False restores the reference's loop shape literally.
        self.inplace_decode = True
        # opt-in: attend straight over the INT8 / INT4 store (quantization/fused_attention.py): no
        # fp16 copy of the cache exists, the model's attention function is kvq_decode_attn
        self.fused_attention = False
        # opt-in, with fused_attention: after two eager steps the decode step (model forward + kvq_decode_step_dev
        # per layer + arg-max) is captured ONCE into a HIP graph and replayed per token (_graph_decode): the
        # reference's Python loop is launch-bound (≈ 3.3 ms of host work per GPT-2 token), the graph is not
        self.graph_decode = False

    # ------------------------------------------------------------------ shared loop machinery

    def _encode(self, prompt: str, truncate: bool) -> torch.Tensor:
        if truncate:
            enc = self.tokenizer(prompt, return_tensors="pt", truncation=True, max_length=1024)
        else:
            enc = self.tokenizer(prompt, return_tensors="pt")
        return enc.input_ids.to(self.device)

    def _compute_dtype(self, past_kv_tuple) -> torch.dtype:
        """fp16 on "cuda", fp32 otherwise — by the device STRING, like the reference
        (benchmarker.py:452, :515). One deviation: a bf16 model keeps bf16, because the reference's
        fp16 choice makes HF's attention fail there (query bf16 vs cached keys fp16/fp32)."""
        if self.device != "cuda":
            return torch.float32
        return torch.bfloat16 if past_kv_tuple[0][0].dtype == torch.bfloat16 else torch.float16

    def _finish(self, generated: torch.Tensor, input_ids: torch.Tensor) -> Tuple[str, int]:
        n_new = generated.shape[-1] - input_ids.shape[-1]
        return self.tokenizer.decode(generated[0], skip_special_tokens=True), n_new

    def _decode_with_policy(self, input_ids: torch.Tensor, max_new_tokens: int,
                            after_forward: Callable[[tuple], object]) -> torch.Tensor:
        """prefill, then ``max_new_tokens`` single-token forwards; ``after_forward`` turns the
        model's KV (legacy tuple) into the cache handed to the NEXT forward."""
        out = self.model(input_ids=input_ids, use_cache=True)
        logits = out.logits[:, -1, :]
        past = after_forward(to_legacy_tuple(out.past_key_values))
        generated = input_ids.clone()
        for _ in range(max_new_tokens):
            next_token = torch.argmax(logits, dim=-1, keepdim=True)
            generated = torch.cat([generated, next_token], dim=-1)
            out = self.model(input_ids=next_token, use_cache=True, past_key_values=past)
            logits = out.logits[:, -1, :]
            past = after_forward(to_legacy_tuple(out.past_key_values))
        self._last_past = past
        return generated

    # ------------------------------------------------------------------ generation methods

    @torch.no_grad()
    def generate_no_cache(self, prompt: str, max_new_tokens: int = 32) -> Tuple[str, int]:
        """Baseline without a KV cache: the whole sequence is re-fed every step; stops at EOS
        (reference benchmarker.py:63-100)."""
        input_ids = self._encode(prompt, truncate=True)
        generated = input_ids.clone()
        vocab_size = self.model.config.vocab_size
        for _ in range(max_new_tokens):
            logits = self.model(input_ids=generated, use_cache=False).logits[:, -1, :]
            next_token = torch.clamp(torch.argmax(logits, dim=-1, keepdim=True), 0, vocab_size - 1)
            generated = torch.cat([generated, next_token], dim=-1)
            if next_token.item() == self.tokenizer.eos_token_id:
                break
        return self._finish(generated, input_ids)

    @torch.no_grad()
    def generate_with_cache(self, prompt: str, max_new_tokens: int = 32) -> Tuple[str, int]:
        """Standard full KV cache kept as the model's own Cache object
        (reference benchmarker.py:102-153)."""
        input_ids = self._encode(prompt, truncate=True)
        out = self.model(input_ids=input_ids, use_cache=True)
        past = out.past_key_values
        if isinstance(past, tuple):
            past = from_legacy_tuple(past)
        logits = out.logits[:, -1, :]
        generated = input_ids.clone()
        vocab_size = self.model.config.vocab_size
        for _ in range(max_new_tokens):
            next_token = torch.clamp(torch.argmax(logits, dim=-1, keepdim=True), 0, vocab_size - 1)
            generated = torch.cat([generated, next_token], dim=-1)
            out = self.model(input_ids=next_token, use_cache=True, past_key_values=past)
            past = out.past_key_values
            logits = out.logits[:, -1, :]
        return self._finish(generated, input_ids)

    @torch.no_grad()
    def generate_with_sliding_window(self, prompt: str, max_new_tokens: int = 32,
                                     window_size: int = 256) -> Tuple[str, int]:
        """Keep only the last ``window_size`` positions after every forward
        (reference benchmarker.py:155-211)."""
        input_ids = self._encode(prompt, truncate=False)
        generated = self._decode_with_policy(
            input_ids, max_new_tokens,
            lambda kv: from_legacy_tuple(trim_kv_sliding_window(kv, window_size)))
        return self._finish(generated, input_ids)

    @torch.no_grad()
    def generate_with_quantized_kv(self, prompt: str, max_new_tokens: int = 32,
                                   mode: str = "int8") -> Tuple[str, int, float]:
        """Decode with the KV cache stored quantised (int8 / int4 / mixed); returns
        ``(text, n_new, estimated cache MB)`` (reference benchmarker.py:422-491)."""
        input_ids = self._encode(prompt, truncate=False)
        if self.graph_decode and not self.fused_attention:
            raise RuntimeError("kvq: graph_decode captures the fused-attention decode step; set fused_attention = True")
        if self.fused_attention:
            return self._generate_fused(input_ids, max_new_tokens, mode)
        out = self.model(input_ids=input_ids, use_cache=True)
        logits = out.logits[:, -1, :]
        past_kv_tuple = to_legacy_tuple(out.past_key_values)

        compute_dtype = self._compute_dtype(past_kv_tuple)
        qcache = QuantizedKVCache(n_layers=len(past_kv_tuple), mode=mode, device=self.device,
                                  compute_dtype=compute_dtype)
        qcache.reserve(input_ids.shape[-1] + max_new_tokens)  # decode never reallocates
        qcache.init_from_prompt_past(past_kv_tuple)  # 2 launches (reference: L*T Python iterations)

        generated = input_ids.clone()
        staged = None
        # in place only when the model's KV dtype is the cache's compute dtype (fp16 on the GPU):
        # the staging buffers are then both what the model reads and what gets quantised
        if self.inplace_decode and hf_cache.available() and past_kv_tuple[0][0].dtype == compute_dtype:
            staged = hf_cache.StagedQuantizedCache(qcache)
        for _ in range(max_new_tokens):
            next_token = torch.argmax(logits, dim=-1, keepdim=True)
            generated = torch.cat([generated, next_token], dim=-1)
            if staged is not None:
                # O(1) per step: dequantise the newest token into the staging buffers, the model
                # appends its K/V there in place, then that slot is quantised into the store
                out = self.model(input_ids=next_token, use_cache=True, past_key_values=staged.sync())
                logits = out.logits[:, -1, :]
                staged.commit()
                continue
            past = from_legacy_tuple(qcache.to_past_key_values())  # 2 launches, no host sync
            out = self.model(input_ids=next_token, use_cache=True, past_key_values=past)
            logits = out.logits[:, -1, :]
            qcache.append_from_past(to_legacy_tuple(out.past_key_values))  # 2 launches

        text, n_new = self._finish(generated, input_ids)
        return text, n_new, mb(qcache.estimated_bytes())

    def _generate_fused(self, input_ids: torch.Tensor, max_new_tokens: int, mode: str) -> Tuple[str, int, float]:
        """quant_* decode with the model attending over the quantised store itself: the prompt forward
        quantises its K/V layer by layer, every later forward runs kvq_decode_attn per layer."""
        cfg = self.model.config
        n_layers = getattr(cfg, "num_hidden_layers", None) or cfg.n_layer
        dtype = next(self.model.parameters()).dtype
        fc = fused_attention.FusedQuantizedCache(n_layers, mode=mode, device=self.device, compute_dtype=dtype,
                                                 reserve=input_ids.shape[-1] + max_new_tokens)
        generated = input_ids.clone()
        with fused_attention.fused_attention(self.model, fc) as cache:
            out = self.model(input_ids=input_ids, use_cache=True, past_key_values=cache)
            logits = out.logits[:, -1, :]
            # graph mode keeps a few eager steps (lazy initialisations, workspace and plan set-up), then captures
            n_eager = min(max_new_tokens, 2) if self.graph_decode else max_new_tokens
            for _ in range(n_eager):
                next_token = torch.argmax(logits, dim=-1, keepdim=True)
                generated = torch.cat([generated, next_token], dim=-1)
                out = self.model(input_ids=next_token, use_cache=True, past_key_values=cache)
                logits = out.logits[:, -1, :]
            if max_new_tokens > n_eager:
                generated = torch.cat([generated, self._graph_decode(fc, cache, logits, max_new_tokens - n_eager)], dim=-1)
        text, n_new = self._finish(generated, input_ids)
        return text, n_new, mb(fc.estimated_bytes())

    def _graph_decode(self, fc, cache, logits: torch.Tensor, n_steps: int) -> torch.Tensor:
        """The remaining ``n_steps`` decode steps as ONE captured HIP graph replayed per token: the model's forward
        (every kernel of it), kvq_decode_step_dev per layer (stored-token count in device memory), the arg-max and
        the position / counter updates. The reference's Python loop (benchmarker.py:465-486: ~3.3 ms of host work
        per GPT-2 token here) leaves the critical path. Returns the ``[1, n_steps]`` tokens generated."""
        qc = fc.qcache
        dev = logits.device
        T0 = qc._k.lens[0]
        if T0 + n_steps > qc._k.cap or logits.shape[0] != 1:
            raise RuntimeError("kvq: graphed decode needs batch 1 and a store reserved for every token")
        static_ids = torch.argmax(logits, dim=-1, keepdim=True).clone()
        gen = torch.zeros(1, n_steps, dtype=static_ids.dtype, device=dev)
        pos = torch.full((1, 1), T0, dtype=torch.long, device=dev)
        step = torch.zeros(1, dtype=torch.long, device=dev)
        fc.t_dev = torch.full((1,), T0, dtype=torch.int32, device=dev)
        fc.t_bound = max(1, T0 + n_steps - 1)
        fc.ignore_decode_mask = True  # un-padded single prompt: the materialised causal row is all-true

        def body():
            gen.index_copy_(1, step, static_ids)  # the token fed to this forward is part of the output
            out = self.model(input_ids=static_ids, position_ids=pos, use_cache=True, past_key_values=cache)
            static_ids.copy_(torch.argmax(out.logits[:, -1, :], dim=-1, keepdim=True))
            pos.add_(1)
            step.add_(1)
            fc.t_dev.add_(1)

        def advance(n):  # host-side mirror of what n executed steps did to the stores
            for i in range(len(qc._k.lens)):
                qc._k.lens[i] += n
                qc._v.lens[i] += n

        try:
            done = 0
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):  # warm-up of the device-T path, executed for real
                for _ in range(min(2, n_steps)):
                    body()
                    done += 1
            torch.cuda.current_stream(dev).wait_stream(side)
            advance(done)
            if done < n_steps:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):  # capture only: nothing executes, host-side counts untouched
                    body()
                for _ in range(n_steps - done):
                    graph.replay()
                advance(n_steps - done)
            torch.cuda.current_stream(dev).synchronize()
        finally:
            fc.t_dev = None
            fc.ignore_decode_mask = False
        return gen

    @torch.no_grad()
    def generate_with_chunked_cache(self, prompt: str, max_new_tokens: int = 32, chunk_size: int = 64,
                                    keep_last: int = 256) -> Tuple[str, int, float]:
        """Chunk-summary cache: older KV mean-pooled per chunk, last ``keep_last`` exact; the
        summary is re-applied to the already summarised cache every step, as the reference does
        (benchmarker.py:570-639). Returns ``(text, n_new, cache MB)``."""
        input_ids = self._encode(prompt, truncate=False)
        generated = self._decode_with_policy(
            input_ids, max_new_tokens,
            lambda kv: from_legacy_tuple(chunk_summarize_kv(kv, chunk_size=chunk_size, keep_last=keep_last)))
        est_bytes = 0
        for k, v in to_legacy_tuple(self._last_past):
            est_bytes += k.numel() * k.element_size() + v.numel() * v.element_size()
        text, n_new = self._finish(generated, input_ids)
        return text, n_new, est_bytes / (1024**2)

    @torch.no_grad()
    def generate_with_prefix_window(self, prompt: str, max_new_tokens: int = 32, window_size: int = 256,
                                    prefix_len: int = 32) -> Tuple[str, int]:
        """Keep the first ``prefix_len`` + the last ``window_size`` tokens (reference
        benchmarker.py:213-268; prompt truncated to 1024 tokens like the reference)."""
        input_ids = self._encode(prompt, truncate=True)
        generated = self._decode_with_policy(
            input_ids, max_new_tokens,
            lambda kv: from_legacy_tuple(trim_kv_prefix_window(kv, prefix_len=prefix_len, window_size=window_size)))
        return self._finish(generated, input_ids)

    @torch.no_grad()
    def generate_with_strided_cache(self, prompt: str, max_new_tokens: int = 32, window_size: int = 256,
                                    stride: int = 4, prefix_len: int = 0) -> Tuple[str, int]:
        """Dense tail + every ``stride``-th older token (reference benchmarker.py:270-318)."""
        input_ids = self._encode(prompt, truncate=True)
        generated = self._decode_with_policy(
            input_ids, max_new_tokens,
            lambda kv: from_legacy_tuple(trim_kv_strided(kv, window_size=window_size, stride=stride,
                                                         prefix_len=prefix_len)))
        return self._finish(generated, input_ids)

    @torch.no_grad()
    def generate_with_block_cache(self, prompt: str, max_new_tokens: int = 32, window_size: int = 256,
                                  block_size: int = 64, keep_per_block: int = 8, prefix_len: int = 0) -> Tuple[str, int]:
        """Dense tail + the last ``keep_per_block`` tokens of every older block (reference
        benchmarker.py:320-370)."""
        input_ids = self._encode(prompt, truncate=True)
        generated = self._decode_with_policy(
            input_ids, max_new_tokens,
            lambda kv: from_legacy_tuple(trim_kv_block_old(kv, window_size=window_size, block_size=block_size,
                                                           keep_per_block=keep_per_block, prefix_len=prefix_len)))
        return self._finish(generated, input_ids)

    @torch.no_grad()
    def generate_with_budget_cache(self, prompt: str, max_new_tokens: int = 32, window_size: int = 256,
                                   old_budget: int = 64, prefix_len: int = 0) -> Tuple[str, int]:
        """Dense tail + a fixed budget of uniformly sampled older tokens (reference
        benchmarker.py:372-420)."""
        input_ids = self._encode(prompt, truncate=True)
        generated = self._decode_with_policy(
            input_ids, max_new_tokens,
            lambda kv: from_legacy_tuple(trim_kv_budget_old(kv, window_size=window_size, old_budget=old_budget,
                                                            prefix_len=prefix_len)))
        return self._finish(generated, input_ids)

    @torch.no_grad()
    def generate_with_paged_attention(self, prompt: str, max_new_tokens: int = 32,
                                      block_size: int = 64) -> Tuple[str, int, float, float, int]:
        """Simulated paged attention: KV kept in fixed-size blocks, stitched before every forward;
        returns ``(text, n_new, alloc_mb, used_mb, num_blocks)`` (reference benchmarker.py:493-568)."""
        input_ids = self._encode(prompt, truncate=False)
        out = self.model(input_ids=input_ids, use_cache=True)
        logits = out.logits[:, -1, :]
        past_kv_tuple = to_legacy_tuple(out.past_key_values)
        dtype = self._compute_dtype(past_kv_tuple)
        paged = [PagedKVCache(block_size=block_size, device=self.device, dtype=dtype) for _ in past_kv_tuple]
        for layer_cache, (k, v) in zip(paged, past_kv_tuple):
            layer_cache.extend(k, v)  # == T single-token appends (reference :524-526)
        generated = input_ids.clone()
        for _ in range(max_new_tokens):
            next_token = torch.argmax(logits, dim=-1, keepdim=True)
            generated = torch.cat([generated, next_token], dim=-1)
            past = from_legacy_tuple(tuple(layer_cache.get_kv() for layer_cache in paged))
            out = self.model(input_ids=next_token, use_cache=True, past_key_values=past)
            logits = out.logits[:, -1, :]
            for layer_cache, (k, v) in zip(paged, to_legacy_tuple(out.past_key_values)):
                layer_cache.append(k[:, :, -1:, :], v[:, :, -1:, :])
        text, n_new = self._finish(generated, input_ids)
        alloc = sum(lc.allocated_bytes() for lc in paged)
        used = sum(lc.used_bytes() for lc in paged)
        return text, n_new, mb(alloc), mb(used), sum(lc.num_blocks() for lc in paged)

    # ------------------------------------------------------------------ benchmarking

    def benchmark_method(self, prompts: list, method: str = "no_cache", max_new_tokens: int = 32,
                         window_size: int = 256, block_size: int = 64, chunk_size: int = 64,
                         keep_last: int = 256, mode: str = "int8", prefix_len: int = 32, stride: int = 4,
                         keep_per_block: int = 8, old_budget: int = 64) -> dict:
        """Run ``method`` over ``prompts`` (serially, B = 1 each, like the reference) and return
        the reference's result dict (benchmarker.py:811-832)."""
        assert method in VALID_METHODS, f"Invalid method: {method}"

        reset_gpu_peak(self.device)
        start_cpu = get_cpu_mem_mb()
        on_gpu = self.device == "cuda"
        if on_gpu:
            torch.cuda.synchronize()
            start_event = torch.cuda.Event(enable_timing=True)
            end_event = torch.cuda.Event(enable_timing=True)
            start_event.record()
        else:
            t0 = time.time()

        total_new_tokens = 0
        est_cache_mbs = []
        for prompt in prompts:
            if method == "no_cache":
                _, n_new = self.generate_no_cache(prompt, max_new_tokens)
                est = 0.0
            elif method == "full_cache":
                _, n_new = self.generate_with_cache(prompt, max_new_tokens)
                est = float("nan")
            elif method == "sliding_window":
                _, n_new = self.generate_with_sliding_window(prompt, max_new_tokens, window_size=window_size)
                est = float("nan")
            elif method in ("quant_int8", "quant_int4", "quant_mixed"):
                # the mode comes from the method name; the `mode` argument is ignored (reference :719-735)
                _, n_new, est = self.generate_with_quantized_kv(prompt, max_new_tokens, mode=method[len("quant_"):])
            elif method == "chunked_cache":
                _, n_new, est = self.generate_with_chunked_cache(prompt, max_new_tokens, chunk_size=chunk_size,
                                                                 keep_last=keep_last)
            elif method == "paged_attention":
                _, n_new, est, _used, _nb = self.generate_with_paged_attention(prompt, max_new_tokens,
                                                                               block_size=block_size)
            elif method == "prefix_window":
                _, n_new = self.generate_with_prefix_window(prompt, max_new_tokens=max_new_tokens,
                                                            window_size=window_size, prefix_len=prefix_len)
                est = float("nan")
            elif method == "strided_cache":
                _, n_new = self.generate_with_strided_cache(prompt, max_new_tokens=max_new_tokens,
                                                            window_size=window_size, stride=stride, prefix_len=prefix_len)
                est = float("nan")
            elif method == "block_cache":
                _, n_new = self.generate_with_block_cache(prompt, max_new_tokens=max_new_tokens,
                                                          window_size=window_size, block_size=block_size,
                                                          keep_per_block=keep_per_block, prefix_len=prefix_len)
                est = float("nan")
            else:  # budget_cache
                _, n_new = self.generate_with_budget_cache(prompt, max_new_tokens=max_new_tokens,
                                                           window_size=window_size, old_budget=old_budget,
                                                           prefix_len=prefix_len)
                est = float("nan")
            est_cache_mbs.append(est)
            total_new_tokens += n_new

        if on_gpu:
            end_event.record()
            torch.cuda.synchronize()
            elapsed = start_event.elapsed_time(end_event) / 1000.0
        else:
            elapsed = time.time() - t0

        finite = [x for x in est_cache_mbs if isinstance(x, float) and not math.isnan(x)]
        windowed = ("sliding_window", "prefix_window", "strided_cache", "block_cache", "budget_cache")
        sparse = ("prefix_window", "strided_cache", "block_cache", "budget_cache")
        return {
            "method": method,
            "elapsed_sec": elapsed,
            "total_new_tokens": total_new_tokens,
            "tokens_per_sec": total_new_tokens / elapsed if elapsed > 0 else float("inf"),
            "cpu_mem_used_mb": get_cpu_mem_mb() - start_cpu,
            "gpu_peak_mb": get_gpu_peak_mb(self.device),
            "window_size": window_size if method in windowed else None,
            "block_size": block_size if method == "paged_attention" else None,
            "chunk_size": chunk_size if method == "chunked_cache" else None,
            "est_kv_cache_mb_avg": sum(finite) / len(finite) if finite else float("nan"),
            "prefix_len": prefix_len if method in sparse else None,
            "stride": stride if method == "strided_cache" else None,
            "keep_per_block": keep_per_block if method == "block_cache" else None,
            "old_budget": old_budget if method == "budget_cache" else None,
        }
