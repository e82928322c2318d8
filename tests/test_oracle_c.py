"""Pins the C oracle (oracle/kvq_oracle.c — checker + cpu_baseline leg) to the numpy oracle and,
through the goldens, to the reference. CPU only."""
import os

import numpy as np
import pytest

from oracle import c_oracle as C
from oracle import kvq_oracle as O
from tests.conftest import ROOT
from tests.util import seeded_kv


def test_half_conversion_exhaustive():
    lib = C.load()
    allh = np.arange(65536, dtype=np.uint16)
    f = allh.view(np.float16).astype(np.float32)
    for h in range(0, 65536, 1):
        if (h & 0x7C00) == 0x7C00 and (h & 0x3FF):
            continue  # NaN payloads are not on the path
        assert lib.kvq_oracle_h2f(h) == f[h] or (np.isinf(f[h]) and np.isinf(lib.kvq_oracle_h2f(h)))
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1e-8, 1e-6, 1e-4, 1.0, 300.0, 7e4)])
    xs = np.concatenate([xs, np.array([65504.0, 65519.9, 65520.0, 2.0**-25, 2.0**-25 * 1.0001, 2.0**-24 * 1.5, 0.0, -0.0], np.float32)])
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([lib.kvq_oracle_f2h(float(v)) for v in xs], dtype=np.uint16)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("shape", [(2, 1, 8, 9, 128), (2, 2, 3, 7, 5), (1, 1, 12, 33, 64)])
@pytest.mark.parametrize("dtype", ["f16", "f32"])
@pytest.mark.parametrize("kind", ["int8", "int4"])
@pytest.mark.parametrize("dist", ["normal", "heavy", "tiny"])
def test_c_quant_dequant_equals_numpy_oracle(shape, dtype, kind, dist):
    x = seeded_kv(shape, dtype, 11, dist)
    q_ref, _, s32_ref = O.quantize_tokens(x, kind)
    q, sc = C.quantize_tokens(x, kind)
    assert np.array_equal(q, q_ref) and np.array_equal(sc.view(np.uint32), s32_ref.view(np.uint32))
    for od in ("f16", "f32"):
        d = C.dequantize_tokens(q, sc, kind, shape[-1], od)
        assert np.array_equal(d.view(np.uint8), O.dequantize_tokens(q_ref, s32_ref, kind, shape[-1], od).view(np.uint8))


@pytest.mark.parametrize("cname", ["tiny", "gpt2ish", "llamaish", "odd"])
@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_c_oracle_vs_reference_goldens(g5, cname, dtype):
    kv = g5[f"{cname}.{dtype}.kv"]
    D = kv.shape[-1]
    for kvi, kind in enumerate(("int8", "int4")):
        q, sc = C.quantize_tokens(kv[:, kvi], kind)
        key = f"{cname}.{dtype}.mixed"
        assert np.array_equal(q, g5[key + (".kq" if kvi == 0 else ".vq")])
        d = C.dequantize_tokens(q, sc, kind, D, dtype)
        assert np.array_equal(d.view(np.uint8), g5[key + ".deq"][:, kvi].view(np.uint8))


@pytest.mark.parametrize("dtype", ["f16", "f32"])
@pytest.mark.parametrize("T,chunk,keep", [(40, 8, 8), (45, 8, 8), (6, 8, 8), (33, 4, 0), (300, 64, 16), (19, 64, 3)])
def test_c_chunk_summary_equals_numpy_oracle(g6, dtype, T, chunk, keep):
    x = g6[f"chunk.{dtype}.T{T}.c{chunk}.k{keep}.x"]
    assert np.array_equal(C.chunk_summarize(x, chunk, keep).view(np.uint8), O.chunk_summarize_kv(x, chunk, keep).view(np.uint8))


@pytest.mark.parametrize("kind", ["int8", "int4"])
def test_literal_loop_equals_numpy_oracle(kind):
    """The per-slice torch-CPU restatement (bench.py's 'literal' CPU leg) agrees with the oracle."""
    import torch
    from oracle import literal_loop as LL
    x = seeded_kv((1, 2, 3, 9, 16), "f16", 5, "heavy")
    q_ref, _, s32 = O.quantize_tokens(x, kind)
    qs, scales = LL.quantize_slices(torch.from_numpy(x[0]), kind)
    assert np.array_equal(torch.cat(qs, dim=2).numpy(), q_ref[0])
    out = LL.dequantize_slices(qs, scales, kind, 16, torch.float16)
    assert np.array_equal(out.numpy().view(np.uint8), O.dequantize_tokens(q_ref, s32, kind, 16, "f16")[0].view(np.uint8))


@pytest.mark.parametrize("kind", ["int8", "int4"])
@pytest.mark.parametrize("shape,dtype", [((2, 2, 3, 9, 16), "f16"), ((1, 1, 8, 5, 128), "f32"), ((2, 2, 3, 7, 5), "f16")])
def test_vectorised_torch_equals_numpy_oracle(kind, shape, dtype):
    """The whole-tensor torch-CPU restatement (bench.py's 'vectorised' CPU leg) agrees with the oracle."""
    import torch
    from oracle import vectorised_torch as VT
    x = seeded_kv(shape, dtype, 7, "heavy")
    q_ref, st_ref, s32 = O.quantize_tokens(x, kind)
    q, sc = VT.quantize_tokens(torch.from_numpy(x), kind)
    assert np.array_equal(q.numpy().view(np.uint8), q_ref.view(np.uint8))
    assert np.array_equal(sc.float().numpy().view(np.uint32), s32.view(np.uint32))
    for od, td in (("f16", torch.float16), ("f32", torch.float32)):
        out = VT.dequantize_tokens(q, sc, kind, shape[-1], td)
        assert np.array_equal(out.numpy().view(np.uint8), O.dequantize_tokens(q_ref, s32, kind, shape[-1], od).view(np.uint8))


@pytest.mark.parametrize("threads", [2, 3, 7])
def test_c_oracle_threaded_entry_points_equal_the_scalar_ones(threads):
    """the `_mt` entry points cut the same scalar loops into contiguous item ranges over pthreads (bench.py's all-core CPU
    baseline): bit-identical to one thread, including ranges that do not divide evenly and more threads than items"""
    rng = np.random.default_rng(11)
    for shape in ((3, 2, 3, 17, 40), (1, 1, 1, 2, 8), (2, 1, 4, 5, 33)):
        x = rng.standard_normal(shape, dtype=np.float32).astype(np.float16)
        for kind in ("int8", "int4"):
            q1, s1 = C.quantize_tokens(x, kind)
            qn, sn = C.quantize_tokens(x, kind, threads=threads)
            assert np.array_equal(q1, qn) and np.array_equal(s1.view(np.uint32), sn.view(np.uint32))
            d1 = C.dequantize_tokens(q1, s1, kind, shape[-1], "f16")
            dn = C.dequantize_tokens(q1, s1, kind, shape[-1], "f16", threads=threads)
            assert np.array_equal(d1.view(np.uint16), dn.view(np.uint16))
        for chunk, keep in ((4, 3), (64, 256), (5, 0)):
            assert np.array_equal(C.chunk_summarize(x[0], chunk, keep).view(np.uint16), C.chunk_summarize(x[0], chunk, keep, threads=threads).view(np.uint16))


def test_c_oracle_is_clean_under_sanitizers():
    """`make -C oracle san`: kvq_oracle.c under AddressSanitizer + UBSan and under ThreadSanitizer, around
    oracle/san_driver.c — every entry point, 1 thread against 2 / 3 / 8 / 37, ragged and empty shapes, exact-size heap
    buffers. A report aborts the driver (non-zero exit); threaded results must equal the single-threaded bytes.
    (GPU sanitizers are not available on the pool: the CPU build is where they run.)"""
    import subprocess
    odir = os.path.join(ROOT, "oracle")
    build = subprocess.run(["make", "-C", odir, "san"], capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stdout[-1500:] + build.stderr[-1500:]
    for exe, env in (("san_asan", {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "halt_on_error=1"}),
                     ("san_tsan", {"TSAN_OPTIONS": "halt_on_error=1"})):
        run = subprocess.run([os.path.join(odir, exe)], capture_output=True, text=True, timeout=300, env={**os.environ, **env})
        if "unexpected memory mapping" in run.stderr:  # the sanitizer runtime against this kernel's address-space randomisation, not a finding
            run = subprocess.run(["setarch", "-R", os.path.join(odir, exe)], capture_output=True, text=True, timeout=300, env={**os.environ, **env})
        assert run.returncode == 0 and "0 mismatches" in run.stdout, (exe, run.stdout[-800:], run.stderr[-3000:])
        assert "Sanitizer" not in run.stderr, (exe, run.stderr[-3000:])
