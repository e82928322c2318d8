"""Configuration surface of the hot path.

Mirrors the reference dataclasses field for field (reference src/core/config.py:10-84) so that
user code constructing ``Config`` / ``QuantizationConfig`` / ``CacheConfig`` / ``BenchmarkConfig``
keeps working. As in the reference, only ``Config`` is consumed by the examples; the real knobs
of a run are the keyword arguments of ``KVCacheBenchmarker.benchmark_method``.
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field
from typing import List, Literal

import torch


def _default_device() -> str:
    return "cuda" if torch.cuda.is_available() else "cpu"


def _default_dtype() -> torch.dtype:
    return torch.float16 if torch.cuda.is_available() else torch.float32


@dataclass
class Config:
    """Run-level settings (reference config.py:10-37). Constructing one seeds python's and
    torch's RNGs with ``seed`` (reference :32-37)."""

    model_name: str = "gpt2"
    device: str = field(default_factory=_default_device)
    dtype: torch.dtype = field(default_factory=_default_dtype)
    seed: int = 42
    max_new_tokens: int = 64
    batch_size: int = 1

    def __post_init__(self) -> None:
        random.seed(self.seed)
        torch.manual_seed(self.seed)
        if self.device == "cuda" and torch.cuda.is_available():
            torch.cuda.manual_seed_all(self.seed)


@dataclass
class QuantizationConfig:
    """KV quantisation settings (reference config.py:40-50)."""

    mode: Literal["int8", "int4", "mixed"] = "int8"
    eps: float = 1e-8


@dataclass
class CacheConfig:
    """Eviction settings (reference config.py:53-67)."""

    window_size: int = 256
    block_size: int = 64
    chunk_size: int = 64
    keep_last: int = 256


@dataclass
class BenchmarkConfig:
    """Sweep lists for the example drivers (reference config.py:70-84)."""

    methods: List[str] = field(default_factory=lambda: ["no_cache", "full_cache", "sliding_window"])
    window_sizes: List[int] = field(default_factory=lambda: [64, 128, 256, 512])
    block_sizes: List[int] = field(default_factory=lambda: [32, 64, 128])
    chunk_sizes: List[int] = field(default_factory=lambda: [32, 64, 128])
