"""Eviction surface (reference src/cache/__init__.py:13-21)."""
from .implementations import (
    PagedKVCache,
    chunk_summarize_kv,
    trim_kv_block_old,
    trim_kv_budget_old,
    trim_kv_prefix_window,
    trim_kv_sliding_window,
    trim_kv_strided,
)

__all__ = [
    "PagedKVCache",
    "trim_kv_sliding_window",
    "chunk_summarize_kv",
    "trim_kv_prefix_window",
    "trim_kv_strided",
    "trim_kv_block_old",
    "trim_kv_budget_old",
]
