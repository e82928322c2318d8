"""Eviction surface: the seven names the reference exports from ``src.cache``
(reference src/cache/__init__.py:13-21), implemented over the HIP kernels in ``implementations``."""
from . import implementations as _impl

__all__ = [
    "trim_kv_sliding_window",   # implementations.py:124-140 in the reference
    "chunk_summarize_kv",       # :295-346
    "trim_kv_prefix_window",    # :143-154
    "trim_kv_strided",          # :157-190
    "trim_kv_block_old",        # :193-245
    "trim_kv_budget_old",       # :248-292
    "PagedKVCache",             # :10-121
]
globals().update({_n: getattr(_impl, _n) for _n in __all__})
