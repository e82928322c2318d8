"""Benchmarking surface on the hot path (reference src/benchmarking/__init__.py:7; the
summarisation / MMLU task harnesses are out of scope)."""
from .benchmarker import KVCacheBenchmarker, from_legacy_tuple, to_legacy_tuple

__all__ = ["KVCacheBenchmarker", "to_legacy_tuple", "from_legacy_tuple"]
