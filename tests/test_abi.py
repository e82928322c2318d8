"""CPU-side checks of the drop-in boundary: libkvq_hip.so loads without a GPU, exports every
symbol include/kvq_hip.h declares, and rejects bad arguments before touching the device.
No compute calls here."""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT

HEADER = os.path.join(ROOT, "include", "kvq_hip.h")


@pytest.fixture(scope="module")
def lib():
    from efficient_llm_inference_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kvq_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    from efficient_llm_inference_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 13
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in kvq_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == declared, "python binding list and header disagree"


def test_version_and_error_string(lib):
    assert lib.kvq_version() == 100
    assert isinstance(lib.kvq_last_error_string(), bytes)


def test_argument_errors_before_any_launch(lib):
    from efficient_llm_inference_amd._lib import KvqDims, KvqStrides, byref
    st, dm = KvqStrides(0, 0, 0, 0), KvqDims(1, 1, 1, 1, 8)
    rc = lib.kvq_dequant_i8_tokens(None, byref(st), None, 0, None, byref(st), 0, byref(dm), None)
    assert rc == -1 and b"NULL" in lib.kvq_last_error_string()
    rc = lib.kvq_dequant_i4_f16_flat(None, 1.0, None, 10, 3, 5, None)  # 10 % 3 != 0
    assert rc == -2
    rc = lib.kvq_dequant_i8_f16_flat(None, 1.0, None, -1, None)
    assert rc == -2
    rc = lib.kvq_dequant_i8_f16_flat(None, 1.0, None, 0, None)  # empty input: ok, no launch
    assert rc == 0
    fake = ctypes.c_void_p(0x1000)
    rc = lib.kvq_quant_i8_tokens(fake, None, byref(st), 7, fake, byref(st), fake, 0, fake, 1e-8, byref(dm), None)
    assert rc == -3  # unknown dtype
    dm0 = KvqDims(0, 1, 1, 1, 8)
    rc = lib.kvq_quant_i8_tokens(fake, None, byref(st), 0, fake, byref(st), fake, 0, fake, 1e-8, byref(dm0), None)
    assert rc == 0  # empty: nothing launched
    rc = lib.kvq_window_compact(fake, None, byref(st), fake, byref(st), 3, 4, byref(dm), None)
    assert rc == -2  # elem_size must be 2 or 4
    rc = lib.kvq_chunk_meanpool(fake, None, byref(st), fake, byref(st), 0, 0, 4, byref(dm), None)
    assert rc == -2  # chunk_size must be > 0


def test_a_failing_call_disarms_the_timing_pair(lib):
    """kvq_time_next_launch arms two events for the thread's NEXT launch. A call that fails validation never launches: the
    library disarms the pair itself, so no later launch — from any entry point — can be timed by mistake."""
    from efficient_llm_inference_amd._lib import KvqDims, KvqStrides, byref
    fake = ctypes.c_void_p(0x1000)  # never dereferenced: nothing launches in this test
    assert lib.kvq_timing_armed() == 0
    assert lib.kvq_time_next_launch(fake, fake) == 0 and lib.kvq_timing_armed() == 1
    st, dm = KvqStrides(0, 0, 0, 0), KvqDims(1, 1, 1, 1, 8)
    assert lib.kvq_dequant_i8_tokens(None, byref(st), None, 0, None, byref(st), 0, byref(dm), None) == -1
    assert lib.kvq_timing_armed() == 0
    # a call that succeeds without launching (empty input) leaves the pair armed; the caller's own disarm clears it
    assert lib.kvq_time_next_launch(fake, None) == 0 and lib.kvq_timing_armed() == 1
    assert lib.kvq_dequant_i8_f16_flat(None, 1.0, None, 0, None) == 0 and lib.kvq_timing_armed() == 1
    assert lib.kvq_time_next_launch(None, None) == 0 and lib.kvq_timing_armed() == 0


def test_chunk_summary_len_matches_reference_trajectory(lib):
    from efficient_llm_inference_amd.kernels import chunk_summary_len
    T, lens = 32768, []
    for _ in range(4):
        T = lib.kvq_chunk_summary_len(T, 64, 256)
        assert T == chunk_summary_len(lens[-1] + 1 if lens else 32768, 64, 256)
        lens.append(T)
        T += 1
    assert lens == [764, 264, 257, 257]  # SURVEY §3.3, reference benchmarker.py:610-626
    assert lib.kvq_chunk_summary_len(6, 8, 8) == 6 and lib.kvq_chunk_summary_len(33, 4, 0) == 9


def test_tunables(lib):
    # test knobs route a call to shipped code: settable in every build
    assert lib.kvq_set_tunable(b"quant_force_two_pass", 1) == 0 and lib.kvq_get_tunable(b"quant_force_two_pass") == 1
    assert lib.kvq_set_tunable(b"quant_force_two_pass", 0) == 0
    assert lib.kvq_set_tunable(b"nope", 1) == -2
    # A-B keys select variants only `make ab` builds contain: the default library refuses them (and says why)
    ab = lib.kvq_is_ab_build()
    for key in (b"dequant_variant", b"quant_tpw", b"attn_fused", b"attn_stream_roll", b"quant_xcd_group", b"attn_merge_fast"):
        before = lib.kvq_get_tunable(key)
        rc = lib.kvq_set_tunable(key, before)
        assert rc == (0 if ab else -2), key
        if not ab:
            assert b"A-B key" in lib.kvq_last_error_string()


def test_default_library_is_the_shipped_subset():
    """The default .so holds the shipped instantiations + generic fallbacks only (the A-B variants live in
    lib/ab/libkvq_hip.so): it stays small, and it is not an A-B build unless KVQ_HIP_LIB says so."""
    from efficient_llm_inference_amd import _lib
    if os.environ.get("KVQ_HIP_LIB"):
        pytest.skip("KVQ_HIP_LIB overrides the library")
    assert not _lib.is_ab_build()
    # the property itself, not a byte count pinned to today's build: an A-B key is refused, and where the A-B library is
    # built next to it the default one is well under half its size
    lib = _lib.load()
    assert lib.kvq_set_tunable(b"dequant_variant", 3) == -2 and b"A-B key" in lib.kvq_last_error_string()
    ab_path = os.path.join(os.path.dirname(_lib.LIB_PATH), "ab", "libkvq_hip.so")
    if os.path.exists(ab_path):
        assert 2 * os.path.getsize(_lib.LIB_PATH) < os.path.getsize(ab_path), (os.path.getsize(_lib.LIB_PATH), os.path.getsize(ab_path))


def test_kernel_log_is_empty_without_launches(lib):
    import ctypes
    lib.kvq_kernel_log_clear()
    buf = ctypes.create_string_buffer(64)
    assert lib.kvq_kernel_log(buf, 64) == 0 and buf.value == b""


def test_no_cpu_fallback_in_product():
    """The product package must not import the oracle nor compute on CPU tensors."""
    import torch
    import efficient_llm_inference_amd as E
    pkg = os.path.join(ROOT, "efficient-llm-inference_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
    x = torch.randn(1, 2, 1, 8)
    for fn in (E.quantize_int8_per_tensor, E.quantize_int4_per_tensor_packed):
        with pytest.raises(RuntimeError):
            fn(x)
    with pytest.raises(RuntimeError):
        E.dequantize_int8_per_tensor(torch.zeros(4, dtype=torch.int8), torch.tensor(1.0), torch.float16)
    with pytest.raises(RuntimeError):
        E.chunk_summarize_kv(((torch.randn(1, 1, 40, 8), torch.randn(1, 1, 40, 8)),), 8, 8)
    qc = E.QuantizedKVCache(2, "mixed", device="cpu", compute_dtype=torch.float32)
    with pytest.raises(RuntimeError):
        qc.init_from_prompt_past(((x, x), (x, x)))
    with pytest.raises(AssertionError):
        E.QuantizedKVCache(1, "int2")
    with pytest.raises(ValueError, match="Empty cache"):
        E.QuantizedLayerKV("int8").get_kv()


def test_decode_attn_workspace_cap_covers_every_length():
    """Host-only: the per-T workspace size of kvq_decode_attn is not monotone in T (tokens per split
    grow with the context), so a decode loop sizes its scratch with kvq_decode_attn_workspace_cap; it
    must cover every T up to the capacity."""
    import random

    from efficient_llm_inference_amd import kernels as K
    rng = random.Random(7)
    for _ in range(60):
        B = rng.choice([1, 2, 8, 64])
        Hkv = rng.choice([1, 2, 8, 12, 32])
        Hq = Hkv * rng.choice([1, 2, 4, 8])
        D = rng.choice([32, 64, 128, 256])
        cap = rng.choice([100, 1000, 5000, 40000, 200000])
        capn = K.decode_attn_workspace_cap(B, Hq, Hkv, cap, D)
        for T in [1, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, cap] + [rng.randint(1, cap) for _ in range(40)]:
            if T <= cap:
                assert K.decode_attn_workspace(B, Hq, Hkv, T, D) <= capn, (B, Hq, Hkv, D, cap, T)
