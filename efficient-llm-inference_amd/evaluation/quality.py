"""Quality helpers for judging cache policies on generation output (scope row N4).

Same names and return values as the reference's ``src/evaluation/quality.py`` (:11-150):
``compute_perplexity``, ``compute_sliding_window_nll``, ``text_similarity``,
``token_agreement_rate``. No kernel content of their own: the sliding-window NLL trims its cache
through ``trim_kv_sliding_window`` (the HIP path; GPU models only) and bridges the transformers cache
formats with the benchmarker's shim. ``text_similarity`` / ``token_agreement_rate`` are pinned to the
reference's outputs by ``tests/golden/g8_round2.npz``.
"""
from __future__ import annotations

import math
from difflib import SequenceMatcher
from typing import List, Sequence, Tuple

import torch


def compute_perplexity(model, tokenizer, texts: List[str], device: str = "cuda",
                       max_length: int = 1024) -> Tuple[float, float]:
    """Teacher-forced NLL over ``texts`` (each truncated to ``max_length`` tokens); returns
    ``(average NLL per token, perplexity)``. Token counts weight the per-text mean losses the way
    the reference does (``loss * numel``, reference quality.py:48-52)."""
    model.eval()
    nll_sum, n_tok = 0.0, 0
    with torch.no_grad():
        for text in texts:
            ids = tokenizer(text, return_tensors="pt", truncation=True, max_length=max_length).input_ids.to(device)
            loss = model(input_ids=ids, labels=ids).loss
            nll_sum += loss.item() * ids.numel()
            n_tok += ids.numel()
    avg = nll_sum / n_tok
    return avg, math.exp(avg)


def compute_sliding_window_nll(model, tokenizer, text: str, window_size: int = 256,
                               device: str = "cuda") -> Tuple[float, float]:
    """Token-by-token NLL of ``text`` when the KV cache only ever holds the last ``window_size``
    positions (reference quality.py:60-121); returns ``(average NLL, perplexity)``."""
    from ..benchmarking.benchmarker import from_legacy_tuple, to_legacy_tuple
    from ..cache import trim_kv_sliding_window

    model.eval()
    ids = tokenizer(text, return_tensors="pt").input_ids.to(device)
    nll_sum, n_tok = 0.0, 0
    past = None
    prev = ids[:, :1]
    with torch.no_grad():
        for i in range(1, ids.size(1)):
            out = model(input_ids=prev, use_cache=True, past_key_values=past)
            logits = out.logits[:, -1, :]
            # the trim IS the hot path (kvq_window_compact): like every other entry of the package it runs on
            # the MI355X only and raises KvqError for host tensors — there is no CPU branch
            kv = trim_kv_sliding_window(to_legacy_tuple(out.past_key_values), window_size)
            past = from_legacy_tuple(kv)
            target = ids[:, i]
            nll_sum += -torch.log_softmax(logits.float(), dim=-1).gather(1, target.unsqueeze(1)).item()
            n_tok += 1
            prev = target.unsqueeze(1)
    avg = nll_sum / n_tok
    return avg, math.exp(avg)


def text_similarity(a: str, b: str) -> float:
    """``difflib.SequenceMatcher`` ratio in [0, 1] (reference quality.py:124-134)."""
    return SequenceMatcher(None, a, b).ratio()


def token_agreement_rate(tok_a: Sequence[int], tok_b: Sequence[int]) -> float:
    """Fraction of equal tokens at equal positions over the shorter length; 0.0 when either is
    empty (reference quality.py:137-150)."""
    n = min(len(tok_a), len(tok_b))
    if n == 0:
        return 0.0
    return sum(1 for x, y in zip(tok_a[:n], tok_b[:n]) if x == y) / n
