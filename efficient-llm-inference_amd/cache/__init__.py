"""Eviction surface on the hot path (reference src/cache/__init__.py:13-21; the index-select
variants and PagedKVCache are scope row N3)."""
from .implementations import chunk_summarize_kv, trim_kv_sliding_window

__all__ = ["trim_kv_sliding_window", "chunk_summarize_kv"]
