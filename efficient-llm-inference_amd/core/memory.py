"""Process / device memory probes and byte arithmetic used by the benchmarker's result dict.

API-compatible with the reference's ``src/core/utils.py`` (function names, arguments, units:
MiB = 2**20 bytes) — ``benchmark_method`` reports ``cpu_mem_used_mb`` and ``gpu_peak_mb`` through
them (reference src/benchmarking/benchmarker.py:689-690, :799-800).
"""
from __future__ import annotations

import os
from typing import Optional

import psutil
import torch

MIB = 1 << 20


def _on_gpu(device: str) -> bool:
    return device == "cuda" and torch.cuda.is_available()


def mb(num_bytes: int) -> float:
    """bytes -> MiB (reference utils.py:49-58)"""
    return num_bytes / MIB


def tensor_bytes(tensor: torch.Tensor) -> int:
    """storage footprint of the tensor's elements (reference utils.py:37-46)"""
    return tensor.element_size() * tensor.numel()


def kv_bytes_fp(k: torch.Tensor, v: torch.Tensor) -> int:
    """footprint of one (K, V) pair (reference utils.py:61-71)"""
    return tensor_bytes(k) + tensor_bytes(v)


def get_cpu_mem_mb() -> float:
    """resident set size of this process in MiB (reference utils.py:10-13)"""
    return mb(psutil.Process(os.getpid()).memory_info().rss)


def reset_gpu_peak(device: str = "cuda") -> None:
    """release cached allocator blocks and restart peak tracking (reference utils.py:16-20)"""
    if _on_gpu(device):
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


def get_gpu_peak_mb(device: str = "cuda") -> Optional[float]:
    """peak bytes handed out by torch's allocator since the last reset, in MiB; None when the
    run is not on the GPU (reference utils.py:23-34)"""
    return mb(torch.cuda.max_memory_allocated()) if _on_gpu(device) else None
