"""Offline stand-ins for the reference examples' ``from_pretrained(<hub name>)`` calls
(reference examples/basic_benchmark.py:21-25, examples/quantized_cache.py:21-25): the build and
GPU boxes have no network and no weights, so BASELINE configs 1-2 run on a RANDOM-INIT model of
the named architecture plus a byte-level tokenizer. Token agreement / text are meaningless with
random weights; tokens/sec, KV-cache MB and the cache-policy code path are exactly the real ones.
Pass a local checkpoint directory to ``load_model`` to use real weights instead.
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import torch

ARCH = {  # architectural constants of the named models (nothing is downloaded)
    "gpt2": dict(n_layer=12, n_head=12, n_embd=768, n_positions=1024, vocab_size=50257),
    "gpt2-medium": dict(n_layer=24, n_head=16, n_embd=1024, n_positions=1024, vocab_size=50257),
    "gpt2-tiny": dict(n_layer=2, n_head=4, n_embd=64, n_positions=256, vocab_size=260),
    "gpt2-mini": dict(n_layer=3, n_head=4, n_embd=256, n_positions=512, vocab_size=260),  # head_dim 64
}
LLAMA_ARCH = {  # grouped-query Llama-family shapes (random init): the grouping and head_dim of Llama-3-8B
    "llama-mini": dict(hidden_size=1024, intermediate_size=2048, num_hidden_layers=4, num_attention_heads=8,
                       num_key_value_heads=2, head_dim=128, vocab_size=260, max_position_embeddings=4096),
    "llama-8b-4layers": dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=4, num_attention_heads=32,
                             num_key_value_heads=8, head_dim=128, vocab_size=260, max_position_embeddings=32768),
    # the whole Llama-3-8B architecture (16 GB of random fp16 weights; vocabulary as published, byte tokens used)
    "llama-8b": dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                     num_key_value_heads=8, head_dim=128, vocab_size=128256, max_position_embeddings=32768),
}


class ByteTokenizer:
    """UTF-8 bytes as token ids 0..255, EOS = 256. Implements the slice of the HF tokenizer API
    the benchmarker uses: ``__call__(..., return_tensors="pt", truncation, max_length).input_ids``,
    ``decode(ids, skip_special_tokens)``, ``eos_token_id``."""

    eos_token_id = 256
    vocab_size = 257

    def __call__(self, text, return_tensors="pt", truncation=False, max_length=None, **_):
        ids = list(text.encode("utf-8")) or [self.eos_token_id]
        if truncation and max_length is not None:
            ids = ids[:max_length]
        return SimpleNamespace(input_ids=torch.tensor([ids], dtype=torch.long))

    def decode(self, ids, skip_special_tokens=True):
        ids = ids.tolist() if hasattr(ids, "tolist") else list(ids)
        data = bytes(i for i in ids if 0 <= i < 256)
        return data.decode("utf-8", errors="replace")


class RepeatTokenizer(ByteTokenizer):
    """ByteTokenizer that maps a prompt ``"<N>"`` to N pseudo-random in-vocabulary tokens: gives
    prompts of an exact token length (e.g. 512) for the BASELINE configs."""

    def __init__(self, vocab_size: int = 257, seed: int = 42):
        self._vocab = vocab_size
        self._seed = seed

    def __call__(self, text, return_tensors="pt", truncation=False, max_length=None, **kw):
        if text.startswith("<") and text.endswith(">") and text[1:-1].isdigit():
            n = int(text[1:-1])
            g = torch.Generator().manual_seed(self._seed + n)
            ids = torch.randint(0, self._vocab, (1, n), generator=g)
            if truncation and max_length is not None:
                ids = ids[:, :max_length]
            return SimpleNamespace(input_ids=ids)
        return super().__call__(text, return_tensors, truncation, max_length, **kw)


def load_model(name_or_path: str = "gpt2", device: str = "cuda", dtype: torch.dtype = torch.float16, seed: int = 42):
    """Random-init GPT-2-family model of the named architecture (offline), or a real checkpoint if
    ``name_or_path`` is a local directory. Returns ``(model.eval(), tokenizer)``."""
    from transformers import GPT2Config, GPT2LMHeadModel

    if os.path.isdir(name_or_path):
        from transformers import AutoModelForCausalLM, AutoTokenizer
        tok = AutoTokenizer.from_pretrained(name_or_path, local_files_only=True)
        model = AutoModelForCausalLM.from_pretrained(name_or_path, local_files_only=True, dtype=dtype)
        return model.to(device).eval(), tok
    if name_or_path in LLAMA_ARCH:
        from transformers import LlamaConfig, LlamaForCausalLM
        a = LLAMA_ARCH[name_or_path]
        torch.manual_seed(seed)
        cfg = LlamaConfig(bos_token_id=0, eos_token_id=256, pad_token_id=None, tie_word_embeddings=False, **a)
        if a["num_hidden_layers"] * a["hidden_size"] > 32 * 1024:  # full-size: initialise on the device, in `dtype`
            prev = torch.get_default_dtype()
            torch.set_default_dtype(dtype)
            try:
                with torch.device(device):
                    model = LlamaForCausalLM(cfg).eval()
            finally:
                torch.set_default_dtype(prev)
        else:
            model = LlamaForCausalLM(cfg).to(device=device, dtype=dtype).eval()
        return model, RepeatTokenizer(vocab_size=a["vocab_size"])
    if name_or_path not in ARCH:
        raise ValueError(f"unknown architecture '{name_or_path}' (known: {sorted(list(ARCH) + list(LLAMA_ARCH))}); "
                         "or pass a local path")
    a = ARCH[name_or_path]
    torch.manual_seed(seed)
    cfg = GPT2Config(bos_token_id=0, eos_token_id=min(256, a["vocab_size"] - 1), **a)
    model = GPT2LMHeadModel(cfg).to(device=device, dtype=dtype).eval()
    return model, RepeatTokenizer(vocab_size=min(a["vocab_size"], 50257))
