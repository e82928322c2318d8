"""MI355X-native KV-cache quantize / dequantize / eviction path.

Drop-in for the hot path of AramBughdaryan/Efficient-LLM-Inference (its ``src`` package):
``Config`` / ``QuantizationConfig`` / ``CacheConfig``, the quantisation ops and containers,
``trim_kv_sliding_window`` / ``chunk_summarize_kv`` and ``KVCacheBenchmarker`` keep their names
and signatures; the arithmetic runs in hand-written HIP kernels for gfx950 behind a C ABI
(``include/kvq_hip.h``, ``libkvq_hip.so``). Import as ``efficient_llm_inference_amd``.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401
from .core.config import BenchmarkConfig, CacheConfig, Config, QuantizationConfig  # noqa: F401
from .cache import chunk_summarize_kv, trim_kv_sliding_window  # noqa: F401
from .quantization import (  # noqa: F401
    QuantizedKVCache,
    QuantizedLayerKV,
    dequantize_int4_per_tensor_packed,
    dequantize_int8_per_tensor,
    quantize_int4_per_tensor_packed,
    quantize_int8_per_tensor,
)
from .hip import get_hip_extension  # noqa: F401
from .benchmarking import KVCacheBenchmarker  # noqa: F401

__all__ = [
    "Config", "QuantizationConfig", "CacheConfig", "BenchmarkConfig",
    "QuantizedKVCache", "QuantizedLayerKV",
    "quantize_int8_per_tensor", "quantize_int4_per_tensor_packed",
    "dequantize_int8_per_tensor", "dequantize_int4_per_tensor_packed",
    "trim_kv_sliding_window", "chunk_summarize_kv", "get_hip_extension", "KVCacheBenchmarker", "__version__",
]
