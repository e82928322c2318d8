"""Memory / size helpers used by the benchmarker (reference src/core/utils.py:10-71)."""
from __future__ import annotations

import os
from typing import Optional

import psutil
import torch

_MIB = float(1024**2)


def get_cpu_mem_mb() -> float:
    """Resident set size of this process in MiB (reference utils.py:10-13)."""
    return psutil.Process(os.getpid()).memory_info().rss / _MIB


def reset_gpu_peak(device: str = "cuda") -> None:
    """Drop cached blocks and zero the peak-allocation counter (reference utils.py:16-20)."""
    if device == "cuda" and torch.cuda.is_available():
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


def get_gpu_peak_mb(device: str = "cuda") -> Optional[float]:
    """Peak torch-allocated HBM in MiB, or None off-GPU (reference utils.py:23-34)."""
    if device == "cuda" and torch.cuda.is_available():
        return torch.cuda.max_memory_allocated() / _MIB
    return None


def tensor_bytes(tensor: torch.Tensor) -> int:
    """numel * itemsize (reference utils.py:37-46)."""
    return tensor.numel() * tensor.element_size()


def mb(num_bytes: int) -> float:
    """bytes -> MiB (reference utils.py:49-58)."""
    return num_bytes / _MIB


def kv_bytes_fp(k: torch.Tensor, v: torch.Tensor) -> int:
    """bytes of one (K, V) pair (reference utils.py:61-71)."""
    return tensor_bytes(k) + tensor_bytes(v)
