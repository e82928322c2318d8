"""Configuration dataclasses and memory probes (the names the reference exports from ``src.core``,
reference src/core/__init__.py:13-24)."""
from . import config as _config
from . import memory as _memory

_EXPORTS = {
    _config: ("Config", "QuantizationConfig", "CacheConfig", "BenchmarkConfig"),
    _memory: ("get_cpu_mem_mb", "get_gpu_peak_mb", "reset_gpu_peak", "tensor_bytes", "mb", "kv_bytes_fp"),
}
__all__ = []
for _mod, _names in _EXPORTS.items():
    for _n in _names:
        globals()[_n] = getattr(_mod, _n)
        __all__.append(_n)

