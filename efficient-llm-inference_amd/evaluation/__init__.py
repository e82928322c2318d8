"""Quality metrics on the hot path's callers (reference src/evaluation/quality.py; the ROUGE and
answer-letter evaluators belong to the out-of-scope task harnesses)."""
from .quality import compute_perplexity, compute_sliding_window_nll, text_similarity, token_agreement_rate

__all__ = ["compute_perplexity", "compute_sliding_window_nll", "text_similarity", "token_agreement_rate"]
