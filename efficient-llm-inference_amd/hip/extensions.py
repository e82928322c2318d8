"""The native-plugin boundary, MI355X edition.

The reference exposes ``build_cuda_extension()`` / ``get_cuda_extension()`` returning a pybind11
module ``kvq_ext`` with ``dequant_int8_to_fp16(q, scale)`` and
``dequant_int4_packed_to_fp16(packed, scale, orig_last_dim)`` (reference
src/cuda/extensions.py:12-147). Here the plugin is ``libkvq_hip.so`` (C ABI,
``include/kvq_hip.h``), built ahead of time by hipcc for gfx950 and bound with ctypes; the object
returned below carries the same two method names with the same argument meaning, allocation
behaviour (returns a new fp16 tensor) and error type (RuntimeError).

Differences, all deliberate: launches go to torch's CURRENT stream; the launch status is
checked; indices are 64-bit; a missing library raises instead of returning ``None``.
"""
from __future__ import annotations

import os
import subprocess
from typing import Optional

import torch

from .. import _lib, kernels

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")


class _KvqExt:
    """ctypes-backed stand-in for the reference's ``kvq_ext`` module (extensions.py:116-119)."""

    name = "kvq_hip"

    @staticmethod
    def dequant_int8_to_fp16(q: torch.Tensor, scale: float) -> torch.Tensor:
        return kernels.dequant_i8_flat(q, scale)

    @staticmethod
    def dequant_int4_packed_to_fp16(packed: torch.Tensor, scale: float, orig_last_dim: int) -> torch.Tensor:
        return kernels.dequant_i4_flat(packed, scale, orig_last_dim)

    @staticmethod
    def version() -> int:
        return int(_lib.load().kvq_version())


_ext: Optional[_KvqExt] = None


def build_hip_extension(verbose: bool = True) -> _KvqExt:
    """Compile libkvq_hip.so for gfx950 (``make -C csrc``; hipcc cross-compiles without a GPU)
    and return the plugin object. Reference: build_cuda_extension (extensions.py:12-135)."""
    global _ext
    proc = subprocess.run(["make", "-C", _CSRC], capture_output=True, text=True)
    if verbose:
        print(proc.stdout[-2000:])
    if proc.returncode != 0:
        raise RuntimeError(f"kvq: building libkvq_hip.so failed:\n{proc.stdout[-4000:]}\n{proc.stderr[-4000:]}")
    _lib.load()
    _ext = _KvqExt()
    return _ext


def get_hip_extension() -> _KvqExt:
    """Load the prebuilt plugin (never JIT-builds on the hot path; the reference calls its
    getter on every dequantise, ops.py:83,112). Raises if the library is missing."""
    global _ext
    if _ext is None:
        _lib.load()
        _ext = _KvqExt()
    return _ext


# reference-compatible aliases
build_cuda_extension = build_hip_extension
get_cuda_extension = get_hip_extension
