"""ctypes binding of libkvq_hip.so (C ABI: include/kvq_hip.h).

This is the only door to the arithmetic of the hot path: there is NO CPU / eager fallback.
If the library is missing or a tensor is not on the GPU, the call raises.

The reference's equivalent is ``get_cuda_extension()`` (reference src/cuda/extensions.py:138-147),
a module-global pybind11 module that silently degrades to ``None`` without CUDA; here a
missing library is an error.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading
from ctypes import POINTER, Structure, byref, c_char_p, c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# KVQ_HIP_LIB: load another build of the same ABI instead — the A-B library (lib/ab/libkvq_hip.so, `make -C csrc ab`)
# for `pytest -m ab` and the sweep scripts, or a calibration build. Unset = the shipped library.
LIB_PATH = os.environ.get("KVQ_HIP_LIB") or os.path.join(_HERE, "lib", "libkvq_hip.so")
AB_LIB_PATH = os.path.join(_HERE, "lib", "ab", "libkvq_hip.so")

KVQ_F16, KVQ_BF16, KVQ_F32 = 0, 1, 2
_DTYPE_CODE = {torch.float16: KVQ_F16, torch.bfloat16: KVQ_BF16, torch.float32: KVQ_F32}

# every symbol include/kvq_hip.h declares (tests/test_abi.py checks the .so exports them all)
EXPORTS = (
    "kvq_version",
    "kvq_last_error_string",
    "kvq_dequant_i8_f16_flat",
    "kvq_dequant_i4_f16_flat",
    "kvq_dequant_i8_tokens",
    "kvq_dequant_i4_tokens",
    "kvq_quant_i8_tokens",
    "kvq_quant_i4_tokens",
    "kvq_absmax_tokens",
    "kvq_absmax_tokens_acc",
    "kvq_quant_tokens_from_absmax",
    "kvq_window_compact",
    "kvq_chunk_meanpool",
    "kvq_chunk_summary_len",
    "kvq_gather_tokens",
    "kvq_decode_attn_workspace",
    "kvq_decode_attn_workspace_cap",
    "kvq_decode_attn",
    "kvq_decode_step",
    "kvq_decode_step_dev",
    "kvq_decode_step_layers",
    "kvq_time_next_launch",
    "kvq_timing_armed",
    "kvq_kernel_log_clear",
    "kvq_kernel_log",
    "kvq_set_tunable",
    "kvq_get_tunable",
    "kvq_is_ab_build",
)


class KvqDims(Structure):
    _fields_ = [("G", c_int64), ("B", c_int64), ("H", c_int64), ("T", c_int64), ("D", c_int64)]


class KvqStrides(Structure):
    _fields_ = [("g", c_int64), ("b", c_int64), ("h", c_int64), ("t", c_int64)]


class KvqAttnDims(Structure):
    _fields_ = [("B", c_int64), ("Hq", c_int64), ("Hkv", c_int64), ("T", c_int64), ("D", c_int64)]


class KvqError(RuntimeError):
    """A libkvq_hip.so entry point returned a non-zero code (reference: TORCH_CHECK ->
    RuntimeError, extensions.py:33-35,72,90,93)."""


_lib = None
_lock = threading.Lock()


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DTYPE_CODE[dt]
    except KeyError:
        raise TypeError(f"kvq: unsupported KV dtype {dt} (float16, bfloat16, float32)") from None


def _declare(lib):
    P = c_void_p
    ST, DM = POINTER(KvqStrides), POINTER(KvqDims)
    lib.kvq_version.restype = c_int
    lib.kvq_version.argtypes = []
    lib.kvq_last_error_string.restype = c_char_p
    lib.kvq_last_error_string.argtypes = []
    lib.kvq_dequant_i8_f16_flat.restype = c_int
    lib.kvq_dequant_i8_f16_flat.argtypes = [P, c_float, P, c_int64, P]
    lib.kvq_dequant_i4_f16_flat.restype = c_int
    lib.kvq_dequant_i4_f16_flat.argtypes = [P, c_float, P, c_int64, c_int64, c_int64, P]
    for name in ("kvq_dequant_i8_tokens", "kvq_dequant_i4_tokens"):
        f = getattr(lib, name)
        f.restype = c_int
        f.argtypes = [P, ST, P, c_int64, P, ST, c_int, DM, P]
    for name in ("kvq_quant_i8_tokens", "kvq_quant_i4_tokens"):
        f = getattr(lib, name)
        f.restype = c_int
        f.argtypes = [P, POINTER(c_void_p), ST, c_int, P, ST, P, c_int64, P, c_float, DM, P]
    for name in ("kvq_absmax_tokens", "kvq_absmax_tokens_acc"):
        f = getattr(lib, name)
        f.restype = c_int
        f.argtypes = [P, POINTER(c_void_p), ST, c_int, P, DM, P]
    lib.kvq_quant_tokens_from_absmax.restype = c_int
    lib.kvq_quant_tokens_from_absmax.argtypes = [c_int, P, POINTER(c_void_p), ST, c_int, P, ST, P, c_int64, P, c_float, DM, P]
    lib.kvq_window_compact.restype = c_int
    lib.kvq_window_compact.argtypes = [P, POINTER(c_void_p), ST, P, ST, c_int, c_int64, DM, P]
    lib.kvq_chunk_meanpool.restype = c_int
    lib.kvq_chunk_meanpool.argtypes = [P, POINTER(c_void_p), ST, P, ST, c_int, c_int64, c_int64, DM, P]
    lib.kvq_gather_tokens.restype = c_int
    lib.kvq_gather_tokens.argtypes = [P, POINTER(c_void_p), ST, P, ST, c_int, P, c_int64, DM, P]
    AD = POINTER(KvqAttnDims)
    lib.kvq_decode_attn_workspace.restype = c_int64
    lib.kvq_decode_attn_workspace.argtypes = [AD]
    lib.kvq_decode_attn_workspace_cap.restype = c_int64
    lib.kvq_decode_attn_workspace_cap.argtypes = [AD]
    lib.kvq_decode_attn.restype = c_int
    lib.kvq_decode_attn.argtypes = [P, c_int64, c_int64, P, ST, P, c_int, P, ST, P, c_int, P, c_int64, c_int64,
                                    P, c_int64, c_int64, P, c_int64, c_int64, c_int, c_float, P, c_int64, AD, P]
    lib.kvq_decode_step.restype = c_int
    lib.kvq_decode_step.argtypes = [P, c_int64, c_int64, P, c_int64, c_int64, P, c_int64, c_int64, P, ST, P, c_int,
                                    P, ST, P, c_int, P, c_int64, c_int64, c_int, c_float, c_float, P, c_int64, AD, P]
    lib.kvq_decode_step_dev.restype = c_int
    lib.kvq_decode_step_dev.argtypes = [P, c_int64, c_int64, P, c_int64, c_int64, P, c_int64, c_int64, P, ST, P, c_int,
                                        P, ST, P, c_int, P, c_int64, c_int64, c_int, c_float, c_float, P, c_int64, AD, P, P]
    PP = POINTER(c_void_p)
    lib.kvq_decode_step_layers.restype = c_int
    lib.kvq_decode_step_layers.argtypes = [c_int64, c_int, PP, c_int64, c_int64, PP, c_int64, c_int64, PP, c_int64, c_int64,
                                           PP, ST, PP, c_int, PP, ST, PP, c_int, PP, c_int64, c_int64, c_int, c_float,
                                           c_float, P, c_int64, AD, P]
    lib.kvq_chunk_summary_len.restype = c_int64
    lib.kvq_chunk_summary_len.argtypes = [c_int64, c_int64, c_int64]
    lib.kvq_time_next_launch.restype = c_int
    lib.kvq_time_next_launch.argtypes = [P, P]
    lib.kvq_timing_armed.restype = c_int
    lib.kvq_timing_armed.argtypes = []
    lib.kvq_kernel_log_clear.restype = None
    lib.kvq_kernel_log_clear.argtypes = []
    lib.kvq_kernel_log.restype = c_int64
    lib.kvq_kernel_log.argtypes = [ctypes.c_char_p, c_int64]
    lib.kvq_is_ab_build.restype = c_int
    lib.kvq_is_ab_build.argtypes = []
    lib.kvq_set_tunable.restype = c_int
    lib.kvq_set_tunable.argtypes = [c_char_p, c_int64]
    lib.kvq_get_tunable.restype = c_int64
    lib.kvq_get_tunable.argtypes = [c_char_p]


def load():
    """Load (once) and return the ctypes handle. Raises if the library has not been built:
    run ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C <pkg>/csrc``."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise KvqError(
                        f"kvq: {LIB_PATH} not found — the HIP library is not built "
                        "(make -C efficient-llm-inference_amd/csrc). There is no CPU fallback."
                    )
                lib = ctypes.CDLL(LIB_PATH)
                _declare(lib)
                _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().kvq_last_error_string().decode("utf-8", "replace")
        raise KvqError(f"kvq{(' ' + what) if what else ''}: {msg} (code {rc})")


def current_stream(device) -> c_void_p:
    """torch's current HIP stream for `device` (the reference launches on the legacy default
    stream instead, extensions.py:79,105)."""
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


class device_guard:
    """``with device_guard(dev):`` — ``dev`` is the thread's current HIP device inside the block (restored after). A no-op
    when it already is: one ``torch.cuda.current_device()`` per launch."""
    __slots__ = ("idx", "prev")

    def __init__(self, device):
        device = torch.device(device)
        self.idx = device.index if device.index is not None else torch.cuda.current_device()
        self.prev = self.idx

    def __enter__(self):
        self.prev = torch.cuda.current_device()
        if self.prev != self.idx:
            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev != self.idx:
            torch.cuda.set_device(self.prev)
        return False


def require_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise KvqError(
            f"kvq: {name} must live on the MI355X (got device '{t.device}'). The hot path has no "
            "CPU implementation in this package; the CPU restatement under oracle/ is test-only."
        )


def strides4(t: torch.Tensor) -> KvqStrides:
    """element strides (g,b,h,t) of a 5-D [G,B,H,T,D] tensor whose last dim is contiguous"""
    assert t.dim() == 5
    g, b, h, tt, d = t.stride()
    if t.size(4) > 1 and d != 1:
        raise KvqError("kvq: last dim must be contiguous")
    return KvqStrides(g, b, h, tt)


def dims5(G, B, H, T, D) -> KvqDims:
    return KvqDims(int(G), int(B), int(H), int(T), int(D))


def ptr_array(ptrs):
    arr = (c_void_p * len(ptrs))(*ptrs)
    return arr


@contextlib.contextmanager
def timed_launch(start: "torch.cuda.Event", stop: "torch.cuda.Event"):
    """``with timed_launch(e0, e1): kernels.<one-launch call>(...)`` — the first kernel launched inside the block
    records its own dispatch start / stop timestamps into the two events (kvq_time_next_launch ->
    hipExtLaunchKernelGGL): the kernel's duration as rocprofv3 reports it, no queue gaps, no barrier packets. Both
    events must have been recorded once before (torch creates the HIP event lazily). The pair is disarmed on the way
    out whatever happened inside (a shape error raised before the library was reached must not leave handles armed
    for a later, unrelated launch)."""
    lib = load()
    check(lib.kvq_time_next_launch(c_void_p(start.cuda_event), c_void_p(stop.cuda_event)), "time_next_launch")
    try:
        yield
    finally:
        lib.kvq_time_next_launch(None, None)


def kernel_log_clear() -> None:
    load().kvq_kernel_log_clear()


def kernel_log() -> list:
    """Demangled names of the distinct kernels this thread has launched since :func:`kernel_log_clear`, in first-launch
    order, shortened the way the profiles quote them (no ``void`` / ``kvq::`` / argument list)."""
    buf = ctypes.create_string_buffer(16384)
    load().kvq_kernel_log(buf, len(buf))
    out = []
    for name in buf.value.decode("utf-8", "replace").splitlines():
        name = name.strip()
        if name.startswith("void "):
            name = name[5:]
        depth, cut = 0, len(name)
        for i, ch in enumerate(name):  # drop the trailing "(kvq::Args ...)": the last top-level parenthesis group
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        out.append(name[:cut].replace("kvq::", ""))
    return out


def is_ab_build() -> bool:
    return bool(load().kvq_is_ab_build())


def set_tunable(key: str, value: int) -> None:
    check(load().kvq_set_tunable(key.encode(), int(value)), "set_tunable")


def get_tunable(key: str) -> int:
    return int(load().kvq_get_tunable(key.encode()))


__all__ = [
    "KvqDims", "KvqStrides", "KvqAttnDims", "KvqError", "load", "check", "current_stream", "require_gpu",
    "strides4", "dims5", "ptr_array", "dtype_code", "set_tunable", "get_tunable", "byref", "timed_launch", "kernel_log",
    "kernel_log_clear", "is_ab_build", "AB_LIB_PATH",
    "c_void_p", "LIB_PATH", "EXPORTS", "KVQ_F16", "KVQ_BF16", "KVQ_F32",
]
