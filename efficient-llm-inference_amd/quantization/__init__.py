"""Quantised-KV surface: the six names the reference exports from ``src.quantization``
(reference src/quantization/__init__.py:12-19), implemented over the HIP kernels in ``ops``."""
from . import ops as _ops

__all__ = [
    "QuantizedKVCache",                   # ops.py:293-363 in the reference
    "QuantizedLayerKV",                   # :136-290
    "quantize_int8_per_tensor",           # :10-30
    "quantize_int4_per_tensor_packed",    # :33-65
    "dequantize_int8_per_tensor",         # :68-90
    "dequantize_int4_per_tensor_packed",  # :93-133
]
globals().update({_n: getattr(_ops, _n) for _n in __all__})
