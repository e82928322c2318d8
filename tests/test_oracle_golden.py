"""Pins the CPU oracle (oracle/kvq_oracle.py) to golden vectors captured from the reference
itself (tests/golden/make_golden.py).  CPU only.

Bar: bit-exact for int8 / packed bytes / stored scales / dequantised values;
chunk mean-pool within 1 storage ulp (torch's summation order differs, see oracle docstring).
"""
import numpy as np
import pytest

from oracle import kvq_oracle as O

SLICES = ["gpt2", "gpt2m", "llama", "odd", "one", "b2"]
DTYPES = ["f32", "f16", "bf16"]
DISTS = ["normal", "heavy", "tiny"]


def _dt(dtype):
    return "bf16" if dtype == "bf16" else None


@pytest.mark.parametrize("sname", SLICES)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("dist", DISTS)
def test_quantize_slices_bit_exact(g1, sname, dtype, dist):
    key = f"{sname}.{dtype}.{dist}"
    x = g1[key + ".x"]
    q8, s8 = O.quantize_int8_per_tensor(x, dtype=_dt(dtype))
    assert np.array_equal(q8, g1[key + ".q8"])
    assert np.array_equal(np.asarray(s8).reshape(1).view(np.uint8), g1[key + ".s8"].view(np.uint8))
    p4, s4, last = O.quantize_int4_per_tensor_packed(x, dtype=_dt(dtype))
    assert np.array_equal(p4, g1[key + ".p4"])
    assert np.array_equal(np.asarray(s4).reshape(1).view(np.uint8), g1[key + ".s4"].view(np.uint8))
    assert last == int(g1[key + ".last"][0])


@pytest.mark.parametrize("sname", SLICES)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("od", DTYPES)
def test_dequantize_slices_bit_exact(g1, sname, dtype, od):
    for dist in DISTS:
        key = f"{sname}.{dtype}.{dist}"
        s8 = O.stored_scale_as_f32(g1[key + ".s8"], _dt(dtype))[0]
        s4 = O.stored_scale_as_f32(g1[key + ".s4"], _dt(dtype))[0]
        d8 = O.dequantize_int8_per_tensor(g1[key + ".q8"], s8, od)
        d4 = O.dequantize_int4_per_tensor_packed(g1[key + ".p4"], s4, int(g1[key + ".last"][0]), od)
        assert np.array_equal(d8.view(np.uint8), g1[key + f".dq8.{od}"].view(np.uint8))
        assert np.array_equal(d4.view(np.uint8), g1[key + f".dq4.{od}"].view(np.uint8))


def test_kat_rounding_and_packing(g2):
    p4, s4, last = O.quantize_int4_per_tensor_packed(g2["kat4.x"])
    assert bytes(p4).hex() == "8aa86f1c" == bytes(g2["kat4.p4"]).hex()
    assert float(s4) == 1.0 and last == 8
    d = O.dequantize_int4_per_tensor_packed(p4, 1.0, 8, "f16")
    assert np.array_equal(d, g2["kat4.dq.f16"])
    q8, s8 = O.quantize_int8_per_tensor(g2["kat8.x"])
    assert np.array_equal(q8, g2["kat8.q8"]) and float(s8) == 1.0
    # half-to-even: 0.5->0, 1.5->2, 2.5->2, -0.5->0, -1.5->-2, -2.5->-2, 126.5->126
    assert q8[:7].tolist() == [0, 2, 2, 0, -2, -2, 126]


@pytest.mark.parametrize("dtype", DTYPES)
def test_zero_slice(g2, dtype):
    z = np.zeros((1, 4, 1, 8), dtype=np.uint16 if dtype == "bf16" else {"f32": np.float32, "f16": np.float16}[dtype])
    q8, s8 = O.quantize_int8_per_tensor(z, dtype=_dt(dtype))
    p4, s4, _ = O.quantize_int4_per_tensor_packed(z, dtype=_dt(dtype))
    assert np.array_equal(q8, g2[f"zero.{dtype}.q8"])
    assert np.array_equal(p4, g2[f"zero.{dtype}.p4"]) and (p4 == 0x88).all()
    assert np.array_equal(np.asarray(s8).reshape(1).view(np.uint8), g2[f"zero.{dtype}.s8"].view(np.uint8))
    assert np.array_equal(np.asarray(s4).reshape(1).view(np.uint8), g2[f"zero.{dtype}.s4"].view(np.uint8))
    if dtype == "f16":  # fp16(1e-8) underflows to 0.0 (SURVEY §7 hard parts)
        assert float(np.asarray(s8)) == 0.0


CACHES = ["tiny", "gpt2ish", "llamaish", "odd"]


@pytest.mark.parametrize("cname", CACHES)
@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("mode", ["int8", "int4", "mixed"])
def test_cache_end_to_end_bit_exact(g5, cname, dtype, mode):
    kv = g5[f"{cname}.{dtype}.kv"]  # [L,2,B,H,T,D]
    key = f"{cname}.{dtype}.{mode}"
    L, _, B, H, T, D = kv.shape
    kinds = {"int8": ("int8", "int8"), "int4": ("int4", "int4"), "mixed": ("int8", "int4")}[mode]
    for kvi, kind in enumerate(kinds):
        x = kv[:, kvi]  # [L,B,H,T,D]
        q, stored, s32 = O.quantize_tokens(x, kind)
        assert np.array_equal(q, g5[key + (".kq" if kvi == 0 else ".vq")])
        assert np.array_equal(stored.view(np.uint8), g5[key + ".scales"][:, kvi].view(np.uint8))
        deq = O.dequantize_tokens(q, s32, kind, D, dtype)
        assert np.array_equal(deq.view(np.uint8), g5[key + ".deq"][:, kvi].view(np.uint8))
    itemsize = 4 if dtype == "f32" else 2
    assert O.estimated_bytes(mode, L, B, H, T, D, itemsize) == int(g5[key + ".bytes"][0])


def test_estimated_bytes_survey_figures():
    # gpt2 shape, T=513, fp32 -> 9.065 / 4.556 / 6.810 MB
    mb = lambda m: O.estimated_bytes(m, 12, 1, 12, 513, 64, 4) / 2**20
    assert abs(mb("int8") - 9.065) < 2e-3 and abs(mb("int4") - 4.556) < 2e-3 and abs(mb("mixed") - 6.810) < 2e-3


@pytest.mark.parametrize("dtype", ["f16", "f32"])
@pytest.mark.parametrize("T,W", [(5, 8), (8, 8), (13, 8), (40, 1)])
def test_sliding_window(g6, dtype, T, W):
    x = g6[f"win.{dtype}.T{T}.W{W}.x"]
    k = O.trim_kv_sliding_window(x, W)
    assert np.array_equal(k, g6[f"win.{dtype}.T{T}.W{W}.k"])
    assert k.shape[-2] == min(T, W)


def _ulp_close(a, b, dtype, xmax):
    """|a-b| <= 1 storage ulp of b + the fp32 re-association error of a chunk sum
    (a few fp32 eps of the largest addend: the error of a sum is relative to its addends,
    not to a cancelled result)."""
    if dtype == "bf16":
        a32, b32 = O.bf16_bits_to_f32(a), O.bf16_bits_to_f32(b)
        tol = 2.0**-7
    else:
        a32, b32 = a.astype(np.float32), b.astype(np.float32)
        tol = {"f16": 2.0**-10, "f32": 2.0**-22}[dtype]
    bound = tol * np.abs(b32) + np.float32(8 * 2.0**-24) * np.float32(xmax)
    return np.all(np.abs(a32 - b32) <= bound)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,chunk,keep", [(40, 8, 8), (45, 8, 8), (6, 8, 8), (33, 4, 0), (300, 64, 16), (19, 64, 3)])
def test_chunk_summary(g6, dtype, T, chunk, keep):
    key = f"chunk.{dtype}.T{T}.c{chunk}.k{keep}"
    x, ref = g6[key + ".x"], g6[key + ".k"]
    out = O.chunk_summarize_kv(x, chunk, keep, dtype=_dt(dtype))
    assert out.shape == ref.shape and out.shape[-2] == O.chunk_summary_len(T, chunk, keep)
    n_sum = out.shape[-2] - min(keep, T) if T > keep else 0
    # recent tail: exact copy
    assert np.array_equal(out[..., n_sum:, :], ref[..., n_sum:, :])
    # summaries: 1 storage ulp (torch CPU mean uses a different fp32 summation order)
    xmax = np.abs(O._widen(x, _dt(dtype))).max()
    assert _ulp_close(out[..., :n_sum, :], ref[..., :n_sum, :], dtype, xmax)


def test_chunk_trajectory(g6):
    T, lens = 32768, []
    for _ in range(4):
        T = O.chunk_summary_len(T, 64, 256)
        lens.append(T)
        T += 1
    assert lens == g6["chunk.trajectory"].tolist() == [764, 264, 257, 257]


@pytest.mark.parametrize("cname", ["llamaish", "odd"])
@pytest.mark.parametrize("mode", ["int8", "int4", "mixed"])
def test_cache_bf16_kv_fp16_compute(cname, mode):
    """bf16 KV, fp16 dequantised output (Llama-family dtype with the reference's GPU compute dtype)."""
    from tests.conftest import load_golden
    g = load_golden("g5b_cache_bf16.npz")
    kv = g[f"{cname}.kv"]  # uint16 bf16 bits [L,2,B,H,T,D]
    L, _, B, H, T, D = kv.shape
    kinds = {"int8": ("int8", "int8"), "int4": ("int4", "int4"), "mixed": ("int8", "int4")}[mode]
    for kvi, kind in enumerate(kinds):
        q, stored, s32 = O.quantize_tokens(kv[:, kvi], kind, dtype="bf16")
        assert np.array_equal(stored, g[f"{cname}.{mode}.scales"][:, kvi])
        deq = O.dequantize_tokens(q, s32, kind, D, "f16")
        assert np.array_equal(deq.view(np.uint8), g[f"{cname}.{mode}.deq"][:, kvi].view(np.uint8))
    assert O.estimated_bytes(mode, L, B, H, T, D, 2) == int(g[f"{cname}.{mode}.bytes"][0])
