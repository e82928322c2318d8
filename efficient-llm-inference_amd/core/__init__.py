"""Config + utility surface (reference src/core/__init__.py:13-24)."""
from .config import BenchmarkConfig, CacheConfig, Config, QuantizationConfig
from .utils import get_cpu_mem_mb, get_gpu_peak_mb, kv_bytes_fp, mb, reset_gpu_peak, tensor_bytes

__all__ = [
    "Config", "QuantizationConfig", "CacheConfig", "BenchmarkConfig",
    "get_cpu_mem_mb", "get_gpu_peak_mb", "reset_gpu_peak", "tensor_bytes", "mb", "kv_bytes_fp",
]
