"""HIP plugin boundary (mirrors reference src/cuda/__init__.py)."""
from .extensions import build_cuda_extension, build_hip_extension, get_cuda_extension, get_hip_extension

__all__ = ["build_hip_extension", "get_hip_extension", "build_cuda_extension", "get_cuda_extension"]
