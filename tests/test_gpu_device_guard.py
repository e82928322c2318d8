"""Device guard: the library launches on the calling thread's CURRENT device. The Python layer makes the
tensors' device current for the call (kernels._launch); the C ABI refuses an output buffer that lives on another device
(KVQ_E_DEVICE) instead of letting the wrong GPU dereference it. The cross-device cases need two visible GPUs (skipped on the
one-GPU box, run by the driver's multi-GPU node); the one-device behaviour is checked everywhere."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import kvq_oracle as O
from tests.util import bits, seeded_kv, to_torch

pytestmark = pytest.mark.gpu


def test_device_guard_is_transparent_on_the_current_device():
    from efficient_llm_inference_amd import _lib
    _lib.load()
    cur = torch.cuda.current_device()
    with _lib.device_guard(torch.device("cuda", cur)) as g:
        assert torch.cuda.current_device() == cur and g.prev == cur
    with _lib.device_guard(torch.device("cuda")):  # no index: the current device
        assert torch.cuda.current_device() == cur
    assert torch.cuda.current_device() == cur
    # a normal call is not refused
    x = torch.randn(1, 1, 8, 4, 128, device="cuda").half()
    import efficient_llm_inference_amd as E
    q, s = E.quantize_int8_per_tensor(x[0, :, :, :1])
    assert q.is_cuda


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible GPUs")
def test_tensors_on_another_device_than_the_current_one():
    import efficient_llm_inference_amd as E
    from efficient_llm_inference_amd import _lib
    from efficient_llm_inference_amd import kernels as K
    lib = _lib.load()
    assert torch.cuda.current_device() == 0
    shape = (2, 1, 8, 37, 128)
    x_np = seeded_kv(shape, "f16", 3, "normal")
    x = to_torch(x_np, device="cuda:1")
    # Python layer: the call follows the tensors, results are the oracle's, the caller's current device is untouched
    qc = E.QuantizedKVCache(1, "mixed", device="cuda:1", compute_dtype=torch.float16)
    qc.init_from_prompt_past(((x[0], x[1]),))
    (k, v), = qc.to_past_key_values()
    torch.cuda.synchronize(1)
    assert k.device.index == 1 and torch.cuda.current_device() == 0
    for got, src, kind in ((k, x_np[0:1], "int8"), (v, x_np[1:2], "int4")):
        q, _, s32 = O.quantize_tokens(src, kind)
        assert np.array_equal(bits(got), bits(O.dequantize_tokens(q, s32, kind, shape[-1], "f16")[0]))
    (kw, _), = E.trim_kv_sliding_window(((x[0], x[1]),), 16)
    torch.cuda.synchronize(1)
    assert torch.equal(kw, x[0][:, :, -16:])
    # C ABI, called the wrong way round (current device 0, buffers on device 1, device 1's stream): refused with KVQ_E_DEVICE
    out = torch.empty(1, 1, 8, 37, 128, dtype=torch.float16, device="cuda:1")
    q8 = torch.zeros(1, 1, 8, 37, 128, dtype=torch.int8, device="cuda:1")
    sc = torch.ones(1, 37, device="cuda:1")
    rc = lib.kvq_dequant_i8_tokens(ctypes.c_void_p(q8.data_ptr()), ctypes.byref(_lib.strides4(q8)), ctypes.c_void_p(sc.data_ptr()), 37,
                                   ctypes.c_void_p(out.data_ptr()), ctypes.byref(_lib.strides4(out)), _lib.dtype_code(torch.float16),
                                   ctypes.byref(_lib.dims5(1, 1, 8, 37, 128)), _lib.current_stream(out.device))
    assert rc == -4 and b"device 1" in lib.kvq_last_error_string()
    with torch.cuda.device(1):  # the right way round
        rc = lib.kvq_dequant_i8_tokens(ctypes.c_void_p(q8.data_ptr()), ctypes.byref(_lib.strides4(q8)), ctypes.c_void_p(sc.data_ptr()), 37,
                                       ctypes.c_void_p(out.data_ptr()), ctypes.byref(_lib.strides4(out)), _lib.dtype_code(torch.float16),
                                       ctypes.byref(_lib.dims5(1, 1, 8, 37, 128)), _lib.current_stream(out.device))
        assert rc == 0
    torch.cuda.synchronize(1)
    assert float(out.abs().max()) == 0.0
