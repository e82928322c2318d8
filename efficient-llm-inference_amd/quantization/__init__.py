"""Quantisation surface (reference src/quantization/__init__.py:12-19)."""
from .ops import (
    QuantizedKVCache,
    QuantizedLayerKV,
    dequantize_int4_per_tensor_packed,
    dequantize_int8_per_tensor,
    quantize_int4_per_tensor_packed,
    quantize_int8_per_tensor,
)

__all__ = [
    "quantize_int8_per_tensor",
    "quantize_int4_per_tensor_packed",
    "dequantize_int8_per_tensor",
    "dequantize_int4_per_tensor_packed",
    "QuantizedLayerKV",
    "QuantizedKVCache",
]
