"""ctypes loader for oracle/libkvq_oracle.so (the C restatement; TEST INFRASTRUCTURE ONLY —
see the header of kvq_oracle.c). Used by tests/ and by bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libkvq_oracle.so")
_lib = None
DT = {"f16": 0, "f32": 2}


def load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "kvq_oracle.c")
        if not os.path.exists(_PATH) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_PATH)):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        lib = ctypes.CDLL(_PATH)
        i64, vp, f32 = ctypes.c_int64, ctypes.c_void_p, ctypes.c_float
        lib.kvq_oracle_quant_tokens.restype = None
        lib.kvq_oracle_quant_tokens.argtypes = [vp, ctypes.c_int, ctypes.c_int, i64, i64, i64, i64, i64, f32, vp, vp]
        lib.kvq_oracle_dequant_tokens.restype = None
        lib.kvq_oracle_dequant_tokens.argtypes = [vp, vp, ctypes.c_int, i64, i64, i64, i64, i64, vp, ctypes.c_int]
        lib.kvq_oracle_chunk_summarize.restype = None
        lib.kvq_oracle_chunk_summarize.argtypes = [vp, ctypes.c_int, i64, i64, i64, i64, i64, vp]
        ci = ctypes.c_int
        lib.kvq_oracle_quant_tokens_mt.restype = None
        lib.kvq_oracle_quant_tokens_mt.argtypes = [vp, ci, ci, i64, i64, i64, i64, i64, f32, vp, vp, ci]
        lib.kvq_oracle_dequant_tokens_mt.restype = None
        lib.kvq_oracle_dequant_tokens_mt.argtypes = [vp, vp, ci, i64, i64, i64, i64, i64, vp, ci, ci]
        lib.kvq_oracle_chunk_summarize_mt.restype = None
        lib.kvq_oracle_chunk_summarize_mt.argtypes = [vp, ci, i64, i64, i64, i64, i64, vp, ci]
        lib.kvq_oracle_f2h.restype = ctypes.c_uint16
        lib.kvq_oracle_f2h.argtypes = [f32]
        lib.kvq_oracle_h2f.restype = f32
        lib.kvq_oracle_h2f.argtypes = [ctypes.c_uint16]
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _np_dt(dtype):
    return {"f16": np.float16, "f32": np.float32}[dtype]


def quantize_tokens(x: np.ndarray, kind: str, eps: float = 1e-8, threads: int = 1):
    """x [G,B,H,T,D] float16|float32 -> (q, scales_f32[G,T]). threads > 1: the same scalar loop, the (g, t) slices
    cut into contiguous ranges over that many pthreads (bit-identical: no value crosses a slice)."""
    x = np.ascontiguousarray(x)
    dtype = "f16" if x.dtype == np.float16 else "f32"
    G, B, H, T, D = x.shape
    bits = 8 if kind == "int8" else 4
    Dq = D if bits == 8 else (D + 1) // 2
    q = np.zeros((G, B, H, T, Dq), dtype=np.int8 if bits == 8 else np.uint8)
    sc = np.zeros((G, T), dtype=np.float32)
    load().kvq_oracle_quant_tokens_mt(_p(x), DT[dtype], bits, G, B, H, T, D, eps, _p(q), _p(sc), int(threads))
    return q, sc


def dequantize_tokens(q: np.ndarray, scales_f32: np.ndarray, kind: str, D: int, out_dtype: str = "f16", threads: int = 1):
    q = np.ascontiguousarray(q)
    sc = np.ascontiguousarray(scales_f32, dtype=np.float32)
    G, B, H, T, _ = q.shape
    out = np.empty((G, B, H, T, D), dtype=_np_dt(out_dtype))
    load().kvq_oracle_dequant_tokens_mt(_p(q), _p(sc), 8 if kind == "int8" else 4, G, B, H, T, D, _p(out), DT[out_dtype], int(threads))
    return out


def chunk_summarize(x: np.ndarray, chunk: int, keep_last: int, threads: int = 1):
    """x [..., T, D] float16|float32"""
    x = np.ascontiguousarray(x)
    dtype = "f16" if x.dtype == np.float16 else "f32"
    T, D = x.shape[-2:]
    R = int(np.prod(x.shape[:-2])) if x.ndim > 2 else 1
    keep = min(keep_last, T)
    old = T - keep
    n = (old + chunk - 1) // chunk if old > 0 else 0
    out = np.empty(x.shape[:-2] + (n + keep, D), dtype=x.dtype)
    load().kvq_oracle_chunk_summarize_mt(_p(x), DT[dtype], R, T, D, chunk, keep_last, _p(out), int(threads))
    return out
