"""GPU parity tests: the HIP path (through the C ABI) vs golden vectors captured from the
reference and vs the CPU oracle on seeded inputs.  Run on a real MI355X: ``pytest -m gpu``.

Bar: bit-exact for int8 values, packed nibbles, stored scales AND dequantised values
(compared as raw bytes, so -0.0 != +0.0); chunk mean-pool bit-exact vs the oracle (same
summation order) and within 1 storage ulp of the reference's goldens.
"""
import zlib

import numpy as np
import pytest
import torch

from oracle import kvq_oracle as O
from tests.util import TD, bits, odt, seeded_kv, to_numpy, to_torch, tunables

pytestmark = pytest.mark.gpu

SLICES = ["gpt2", "gpt2m", "llama", "odd", "one", "b2"]
DTYPES = ["f32", "f16", "bf16"]
DISTS = ["normal", "heavy", "tiny"]


@pytest.fixture(scope="module")
def E():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import efficient_llm_inference_amd as pkg
    from efficient_llm_inference_amd import _lib
    _lib.load()  # fail loudly if the HIP library is missing
    return pkg


# ------------------------------------------------------------------ goldens: per-tensor API


@pytest.mark.parametrize("sname", SLICES)
@pytest.mark.parametrize("dtype", DTYPES)
def test_golden_quantize_per_tensor(E, g1, sname, dtype):
    for dist in DISTS:
        key = f"{sname}.{dtype}.{dist}"
        x = to_torch(g1[key + ".x"], dtype)
        q8, s8 = E.quantize_int8_per_tensor(x)
        p4, s4, last = E.quantize_int4_per_tensor_packed(x)
        assert q8.dtype == torch.int8 and p4.dtype == torch.uint8 and s8.dtype == x.dtype and s8.dim() == 0
        assert np.array_equal(to_numpy(q8), g1[key + ".q8"]), key
        assert np.array_equal(to_numpy(p4), g1[key + ".p4"]), key
        assert np.array_equal(bits(s8.reshape(1)), bits(g1[key + ".s8"])), key
        assert np.array_equal(bits(s4.reshape(1)), bits(g1[key + ".s4"])), key
        assert last == int(g1[key + ".last"][0])


@pytest.mark.parametrize("sname", SLICES)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("od", DTYPES)
def test_golden_dequantize_per_tensor(E, g1, sname, dtype, od):
    for dist in DISTS:
        key = f"{sname}.{dtype}.{dist}"
        q8 = to_torch(g1[key + ".q8"])
        p4 = to_torch(g1[key + ".p4"])
        s8 = to_torch(g1[key + ".s8"], dtype).reshape(())
        s4 = to_torch(g1[key + ".s4"], dtype).reshape(())
        d8 = E.dequantize_int8_per_tensor(q8, s8, TD[od])
        d4 = E.dequantize_int4_per_tensor_packed(p4, s4, int(g1[key + ".last"][0]), TD[od])
        assert np.array_equal(bits(d8), bits(g1[key + f".dq8.{od}"])), key
        assert np.array_equal(bits(d4), bits(g1[key + f".dq4.{od}"])), key


def test_golden_kat_and_zero(E, g2):
    x = to_torch(g2["kat4.x"])
    p4, s4, last = E.quantize_int4_per_tensor_packed(x)
    assert bytes(to_numpy(p4)).hex() == "8aa86f1c" and float(s4) == 1.0 and last == 8
    assert np.array_equal(to_numpy(E.dequantize_int4_per_tensor_packed(p4, s4, 8, torch.float16)), g2["kat4.dq.f16"])
    q8, s8 = E.quantize_int8_per_tensor(to_torch(g2["kat8.x"]))
    assert np.array_equal(to_numpy(q8), g2["kat8.q8"]) and float(s8) == 1.0
    for dtype in DTYPES:
        z = torch.zeros(1, 4, 1, 8, dtype=TD[dtype], device="cuda")
        q8, s8 = E.quantize_int8_per_tensor(z)
        p4, s4, _ = E.quantize_int4_per_tensor_packed(z)
        assert np.array_equal(to_numpy(q8), g2[f"zero.{dtype}.q8"])
        assert np.array_equal(to_numpy(p4), g2[f"zero.{dtype}.p4"])
        assert np.array_equal(bits(s8.reshape(1)), bits(g2[f"zero.{dtype}.s8"]))
        assert np.array_equal(bits(s4.reshape(1)), bits(g2[f"zero.{dtype}.s4"]))
        assert np.array_equal(bits(E.dequantize_int8_per_tensor(q8, s8, torch.float16)), bits(g2[f"zero.{dtype}.dq8.f16"]))
        assert np.array_equal(bits(E.dequantize_int4_per_tensor_packed(p4, s4, 8, torch.float16)), bits(g2[f"zero.{dtype}.dq4.f16"]))


# ------------------------------------------------------------------ goldens: the container


@pytest.mark.parametrize("cname", ["tiny", "gpt2ish", "llamaish", "odd"])
@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("mode", ["int8", "int4", "mixed"])
def test_golden_cache_end_to_end(E, g5, cname, dtype, mode):
    kv = to_torch(g5[f"{cname}.{dtype}.kv"])  # [L,2,B,H,T+1,D]
    key = f"{cname}.{dtype}.{mode}"
    L, _, B, H, T1, D = kv.shape
    T = T1 - 1
    qc = E.QuantizedKVCache(n_layers=L, mode=mode, device="cuda", compute_dtype=TD[dtype])
    with pytest.raises(ValueError, match="Empty cache"):
        qc.layers[0].get_kv()
    qc.init_from_prompt_past(tuple((kv[l, 0, :, :, :T], kv[l, 1, :, :, :T]) for l in range(L)))
    qc.append_from_past(tuple((kv[l, 0], kv[l, 1]) for l in range(L)))
    past = qc.to_past_key_values()
    assert len(past) == L and past[0][0].shape == (B, H, T1, D) and past[0][0].dtype == TD[dtype]
    deq = torch.stack([torch.stack([k, v]) for k, v in past])
    assert np.array_equal(bits(deq), bits(g5[key + ".deq"]))
    assert qc.estimated_bytes() == int(g5[key + ".bytes"][0])
    # stored representation: the reference's lists, as views of the persistent buffers
    kq = torch.stack([torch.cat(layer.k_store, dim=2) for layer in qc.layers])
    vq = torch.stack([torch.cat(layer.v_store, dim=2) for layer in qc.layers])
    assert np.array_equal(to_numpy(kq), g5[key + ".kq"]) and np.array_equal(to_numpy(vq), g5[key + ".vq"])
    sc = torch.stack([torch.stack([torch.stack(l.k_scales), torch.stack(l.v_scales)]) for l in qc.layers])
    assert np.array_equal(bits(sc), bits(g5[key + ".scales"]))
    # per-layer API gives the same tensors
    k0, v0 = qc.layers[0].get_kv()
    assert torch.equal(k0, past[0][0]) and torch.equal(v0, past[0][1])


# ------------------------------------------------------------------ oracle: seeded, both paths


CASES = [  # (G, B, H, T, D)
    (4, 1, 8, 300, 128),   # llama slice shape, ragged token tiles
    (3, 1, 12, 129, 64),   # gpt2
    (2, 1, 16, 64, 64),    # gpt2-medium
    (2, 2, 4, 33, 32),     # batch > 1
    (2, 8, 8, 5, 128),     # R*D = 8192 -> 2 tokens per tile
    (1, 64, 8, 3, 128),    # R*D = 65536 > register tile: swept-tile kernel
    (2, 40, 8, 21, 64),    # R*D = 20480: swept tile, several tokens per tile, ragged last tile
    (2, 2, 3, 7, 5),       # odd D: generic kernels
    (1, 1, 2, 4, 24),      # D % 8 == 0 but D/8 not a power of two: division-indexed swept tile
    (3, 1, 8, 131, 96),    # D/8 = 12 (phi / falcon-style heads), ragged last tile
    (2, 2, 5, 70, 80),     # D/8 = 10, batch > 1, R*D*TT spans several sweep steps
    (1, 40, 8, 9, 160),    # D/8 = 20, R*D = 51200: many sweep steps per tile
    (2, 1, 3, 6, 1024),    # D/8 = 128 > one wave: division-indexed swept tile
]


def _quant_via_kernels(E, x_np, dtype, kind, force_two_pass, as_list, tcap_pad=3, direct_stores=False, block=64, tile=1, wide=1, **ab):
    """quantise through the C ABI into a window of a larger store. Test knobs (every build): force_two_pass,
    direct_stores, block (64 | 256: the one-wave or the 256-thread general kernel), tile (1 = the compile-time tile
    kernel where the shape has one), wide (1 = the single-pass 1024-thread tile for 16384 < B*H*D <= 131072).
    **ab: A-B keys (quant_nv, quant_tpw, ...; `pytest -m ab` only)."""
    from efficient_llm_inference_amd import kernels
    G, B, H, T, D = x_np.shape
    x = to_torch(x_np, dtype)
    Dq = kernels.packed_dim(kind, D)
    store = torch.zeros(G, B, H, T + tcap_pad, Dq, dtype=kernels.QDTYPE[kind], device="cuda")
    scales = torch.zeros(G, T + tcap_pad, dtype=torch.float32, device="cuda")
    ws = torch.empty(G * T + 8, dtype=torch.float32, device="cuda")
    with tunables(quant_force_two_pass=int(force_two_pass), quant_direct_stores=int(direct_stores), quant_block=int(block),
                  quant_tile=int(tile), quant_wide=int(wide), **ab):
        src = [x[g] for g in range(G)] if as_list else x
        kernels.quant_tokens(src, store[:, :, :, 1:T + 1], scales[:, 1:T + 1], ws, kind)
    torch.cuda.synchronize()
    # the window [1, T+1) was written; the guard tokens around it must be untouched
    assert int(store[:, :, :, 0].to(torch.int32).abs().sum()) == 0 and int(store[:, :, :, T + 1:].to(torch.int32).abs().sum()) == 0
    assert float(scales[:, 0].abs().sum()) == 0.0 and float(scales[:, T + 1:].abs().sum()) == 0.0
    return store, scales


@pytest.mark.ab
@pytest.mark.parametrize("case", [(4, 1, 8, 300, 128), (2, 1, 8, 64, 128), (1, 1, 8, 1031, 128), (3, 2, 4, 37, 128), (2, 1, 8, 40, 64)])
@pytest.mark.parametrize("tpw", [0, 2, 4, 8])
@pytest.mark.parametrize("kind", ["int8", "int4"])
def test_quant_pipelined_one_wave_tiles(E, case, tpw, kind):
    """quant_tokens_pipe_k (a wave walks `tpw` tiles, next tile's loads ahead of this tile's stores): bit-exact
    q and stored scales for every tiles-per-wave setting, with tile counts that leave a remainder for the
    one-tile kernel and a ragged last tile, list and single-buffer inputs, fp16 and bf16, all three
    distributions (\"tiny\" drives the exact-division fallback and zero stored scales)."""
    G, B, H, T, D = case
    for dtype, dist, as_list in (("f16", "normal", False), ("f16", "heavy", True), ("bf16", "heavy", False), ("f16", "tiny", True)):
        x_np = seeded_kv(case, dtype, seed=zlib.crc32(repr((case, dtype, kind, dist, "pipe")).encode()), dist=dist)
        q_ref, _, s32_ref = O.quantize_tokens(x_np, kind, dtype=odt(dtype))
        store, scales = _quant_via_kernels(E, x_np, dtype, kind, False, as_list, block=64, tile=0, quant_tpw=tpw)
        assert np.array_equal(to_numpy(store[:, :, :, 1:T + 1]), q_ref), (dtype, dist, tpw)
        assert np.array_equal(bits(scales[:, 1:T + 1]), bits(s32_ref)), (dtype, dist, tpw)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", ["int8", "int4"])
def test_oracle_quant_dequant_tokens(E, case, dtype, kind):
    from efficient_llm_inference_amd import kernels
    G, B, H, T, D = case
    # every SHIPPED path a shape can take: the compile-time tile kernel (where the shape has one), the general one-wave
    # kernel, the 256-thread kernel with and without LDS-staged stores, the generic two-pass pair
    for dist, two_pass, as_list, direct, block, tile in (
            ("normal", False, False, False, 256, 0), ("heavy", True, True, False, 256, 0), ("tiny", False, True, False, 256, 0),
            ("heavy", False, False, True, 256, 0), ("heavy", False, True, False, 64, 0), ("heavy", False, False, False, 64, 1),
            ("tiny", False, True, False, 64, 1)):
        x_np = seeded_kv(case, dtype, seed=zlib.crc32(repr((case, dtype, kind, dist)).encode()), dist=dist)
        q_ref, stored_ref, s32_ref = O.quantize_tokens(x_np, kind, dtype=odt(dtype))
        store, scales = _quant_via_kernels(E, x_np, dtype, kind, two_pass, as_list, direct_stores=direct, block=block, tile=tile)
        assert np.array_equal(to_numpy(store[:, :, :, 1:T + 1]), q_ref), (dist, two_pass, block, tile)
        assert np.array_equal(bits(scales[:, 1:T + 1]), bits(s32_ref)), (dist, two_pass, block, tile)
        for od in DTYPES:
            out = torch.zeros(G, B, H, T + 2, D, dtype=TD[od], device="cuda")  # strided output window
            kernels.dequant_tokens(store[:, :, :, 1:T + 1], scales[:, 1:T + 1], out[:, :, :, :T], kind)
            ref = O.dequantize_tokens(q_ref, s32_ref, kind, D, od)
            assert np.array_equal(bits(out[:, :, :, :T]), bits(ref)), (dist, od)
            assert float(out[:, :, :, T:].float().abs().sum()) == 0.0


WIDE_CASES = [  # (G, B, H, T, D): 16384 < B*H*D <= 131072 -> quant_wide_k (two-byte inputs)
    (1, 32, 8, 70, 128),   # 4 tokens per workgroup, ragged last tile
    (2, 64, 8, 33, 128),   # Llama-3-8B batch 64: 2 tokens per workgroup, odd token count
    (1, 128, 8, 5, 128),   # the largest slice the register tile holds: 1 token per workgroup
    (2, 40, 8, 21, 64),    # 320 rows of 64: rounds partly filled
    (1, 10, 5, 9, 512),    # D/8 = 64: a row run of 256 vectors, 4 rows per round
    (3, 64, 8, 1, 128),    # batch-64 decode append
]


@pytest.mark.parametrize("case", WIDE_CASES)
@pytest.mark.parametrize("kind", ["int8", "int4"])
def test_quant_wide_tile(E, case, kind):
    """Batched slices larger than the one-wave tile: the single-pass 1024-thread kernel (default), the split-phase tile
    kernels (quant_wide = 0, head_dim 128) and the swept tile (quant_wide = 0, quant_tile = 0) all give the oracle's bytes
    and stored scales; the kernel log shows which one ran."""
    from efficient_llm_inference_amd import _lib
    G, B, H, T, D = case
    for dtype, dist, as_list in (("f16", "heavy", False), ("bf16", "tiny", True)):
        x_np = seeded_kv(case, dtype, seed=zlib.crc32(repr((case, dtype, kind, dist, "wide")).encode()), dist=dist)
        q_ref, _, s32_ref = O.quantize_tokens(x_np, kind, dtype=odt(dtype))
        for wide, tile in ((1, 1), (0, 1), (0, 0)):
            _lib.kernel_log_clear()
            store, scales = _quant_via_kernels(E, x_np, dtype, kind, False, as_list, wide=wide, tile=tile)
            log = _lib.kernel_log()
            assert np.array_equal(to_numpy(store[:, :, :, 1:T + 1]), q_ref), (dist, wide, tile, log)
            assert np.array_equal(bits(scales[:, 1:T + 1]), bits(s32_ref)), (dist, wide, tile, log)
            if wide:
                assert log and all(k.startswith("quant_wide_k<") for k in log), log
            else:
                assert not any(k.startswith("quant_wide_k<") for k in log), log
                if tile and D == 128:
                    assert any(k.startswith("quant_tile_k<") and k.endswith(", 2>") for k in log), log


@pytest.mark.ab
@pytest.mark.parametrize("case", WIDE_CASES[:5])
@pytest.mark.parametrize("kind", ["int8", "int4"])
def test_quant_wide_tile_512_thread_variant(E, case, kind):
    """A-B library: the wide tile as 512-thread workgroups (two per CU, half the elements each) — same bytes; slices
    above its 65536 elements fall through to the other kernels."""
    G, B, H, T, D = case
    for dtype, dist in (("f16", "heavy"), ("bf16", "tiny")):
        x_np = seeded_kv(case, dtype, seed=zlib.crc32(repr((case, dtype, kind, dist, "wide512")).encode()), dist=dist)
        q_ref, _, s32_ref = O.quantize_tokens(x_np, kind, dtype=odt(dtype))
        store, scales = _quant_via_kernels(E, x_np, dtype, kind, False, False, quant_wide_blk=512)
        assert np.array_equal(to_numpy(store[:, :, :, 1:T + 1]), q_ref) and np.array_equal(bits(scales[:, 1:T + 1]), bits(s32_ref)), (dtype, dist)


@pytest.mark.parametrize("kind", ["int8", "int4"])
def test_quant_wide_tile_strided_batch_rows(E, kind):
    """Every second batch row of a larger tensor (input batch stride != H * head stride) into every second batch row of a
    larger store: the wide kernel serves it when its rows-per-round is a multiple of H (H = 8), other layouts fall through
    to the general kernels (H = 5: 4 rows per round) — same bytes either way."""
    from efficient_llm_inference_amd import _lib, kernels
    for (G, B, H, T, D), served in (((2, 64, 8, 19, 128), True), ((1, 10, 5, 9, 512), False)):
        x_np = seeded_kv((G, 2 * B, H, T, D), "f16", seed=B + H, dist="heavy")
        q_ref, _, s32_ref = O.quantize_tokens(np.ascontiguousarray(x_np[:, ::2]), kind, dtype=odt("f16"))
        x = to_torch(x_np, "f16")
        Dq = kernels.packed_dim(kind, D)
        store = torch.zeros(G, 2 * B, H, T, Dq, dtype=kernels.QDTYPE[kind], device="cuda")
        scales = torch.zeros(G, T, dtype=torch.float32, device="cuda")
        ws = torch.empty(G * T, dtype=torch.float32, device="cuda")
        _lib.kernel_log_clear()
        kernels.quant_tokens(x[:, ::2], store[:, 1::2], scales, ws, kind)
        torch.cuda.synchronize()
        log = _lib.kernel_log()
        assert any(k.startswith("quant_wide_k<") for k in log) == served, log
        assert np.array_equal(to_numpy(store[:, 1::2]), q_ref) and np.array_equal(bits(scales), bits(s32_ref)), log
        assert int(store[:, 0::2].to(torch.int32).abs().sum()) == 0


@pytest.mark.ab
@pytest.mark.parametrize("case", [(4, 1, 8, 300, 128), (3, 1, 12, 129, 64), (2, 1, 16, 64, 64), (2, 2, 4, 33, 32), (2, 8, 8, 5, 128), (2, 1, 8, 77, 64)])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", ["int8", "int4"])
def test_quant_ab_tile_sizes_and_instantiations(E, case, dtype, kind):
    """A-B library only: 128-thread tiles, 2048- / 8192-element one-wave tiles, round 2's GEO128 instantiation with its
    non-temporal on / off variants, the half-wave row runs of the tile kernel at head_dim 64 — all bit-exact."""
    G, B, H, T, D = case
    for dist, as_list, block, ab in (("normal", False, 128, {}), ("heavy", False, 64, {"quant_nv": 4}), ("normal", True, 64, {"quant_nv": 16}),
                                     ("heavy", False, 64, {"quant_geo128": 1}), ("heavy", True, 64, {"quant_geo128": 1, "nt_loads": 0}),
                                     ("tiny", False, 64, {"quant_geo128": 1, "quant_nt_stores": 0}), ("heavy", False, 64, {"quant_no_regmax": 1})):
        x_np = seeded_kv(case, dtype, seed=zlib.crc32(repr((case, dtype, kind, dist, "ab")).encode()), dist=dist)
        q_ref, _, s32_ref = O.quantize_tokens(x_np, kind, dtype=odt(dtype))
        store, scales = _quant_via_kernels(E, x_np, dtype, kind, False, as_list, block=block, tile=0, **ab)
        assert np.array_equal(to_numpy(store[:, :, :, 1:T + 1]), q_ref), (dist, block, ab)
        assert np.array_equal(bits(scales[:, 1:T + 1]), bits(s32_ref)), (dist, block, ab)
    if B * H == 8 and dtype == "f16":  # merged-store tile kernel: 2 / 4 tiles per wave, ragged token counts included
        for tpw, dist in ((2, "heavy"), (4, "tiny"), (2, "normal")):
            x_np = seeded_kv(case, dtype, seed=7 + tpw, dist=dist)
            q_ref, _, s32_ref = O.quantize_tokens(x_np, kind, dtype=odt(dtype))
            store, scales = _quant_via_kernels(E, x_np, dtype, kind, False, tpw == 4, tile=1, quant_tile_tpw=tpw)
            assert np.array_equal(to_numpy(store[:, :, :, 1:T + 1]), q_ref) and np.array_equal(bits(scales[:, 1:T + 1]), bits(s32_ref)), (tpw, dist)
    if D == 64 and dtype != "f32":
        x_np = seeded_kv(case, dtype, seed=5, dist="heavy")
        q_ref, _, s32_ref = O.quantize_tokens(x_np, kind, dtype=odt(dtype))
        store, scales = _quant_via_kernels(E, x_np, dtype, kind, False, False, tile=1, quant_tile_tt=4)
        assert np.array_equal(to_numpy(store[:, :, :, 1:T + 1]), q_ref) and np.array_equal(bits(scales[:, 1:T + 1]), bits(s32_ref))


@pytest.mark.ab
@pytest.mark.parametrize("kind", ["int8", "int4"])
@pytest.mark.parametrize("od", DTYPES)
def test_dequant_all_variants_bit_exact(E, kind, od):
    """Every tuning variant of the fast dequantise kernel gives identical bytes."""
    from efficient_llm_inference_amd import _lib, kernels
    case = (3, 1, 8, 1031, 128)  # ragged: row length not a multiple of any chunk
    x_np = seeded_kv(case, "f16", seed=77, dist="heavy")
    q_ref, _, s32_ref = O.quantize_tokens(x_np, kind)
    ref = O.dequantize_tokens(q_ref, s32_ref, kind, 128, od)
    q, sc = to_torch(q_ref), to_torch(s32_ref)
    try:
        for v in range(36):
            for grid in (0, 7):
                _lib.set_tunable("dequant_variant", v)
                _lib.set_tunable("dequant_grid", grid)
                out = torch.empty(case, dtype=TD[od], device="cuda")
                kernels.dequant_tokens(q, sc, out, kind)
                assert np.array_equal(bits(out), bits(ref)), (v, grid)
    finally:
        _lib.set_tunable("dequant_variant", -1)
        _lib.set_tunable("dequant_grid", 0)


def test_flat_entry_points(E):
    """The reference's own plugin entry points (extensions.py:70-114)."""
    ext = E.get_hip_extension()
    rng = np.random.default_rng(5)
    for n in (1, 7, 768, 1024, 4099):
        q = rng.integers(-127, 128, size=(n,), dtype=np.int8)
        out = ext.dequant_int8_to_fp16(to_torch(q), 0.0123)
        assert out.dtype == torch.float16
        assert np.array_equal(bits(out), bits(O.dequantize_int8_per_tensor(q, np.float32(0.0123), "f16")))
    for shape, orig in (((2, 3, 1, 3), 5), ((1, 8, 1, 64), 128), ((5, 4), 8), ((3, 1), 1)):
        p = rng.integers(0, 256, size=shape, dtype=np.uint8)
        out = ext.dequant_int4_packed_to_fp16(to_torch(p), 0.37, orig)
        assert out.shape[-1] == 2 * shape[-1]
        assert np.array_equal(bits(out), bits(O.dequant_int4_packed_kernel_full(p, np.float32(0.37), orig)))
    # large buffers take the one-wave-per-chunk kernel (>= 1 Mi elements): whole chunks, a ragged last chunk, one group over
    for n in (1 << 20, (1 << 20) + 8, 3 * (1 << 20) + 1024 * 5 + 8):
        q = rng.integers(-127, 128, size=(n,), dtype=np.int8)
        out = ext.dequant_int8_to_fp16(to_torch(q), 0.0123)
        assert np.array_equal(bits(out), bits(O.dequantize_int8_per_tensor(q, np.float32(0.0123), "f16")))
        p = rng.integers(0, 256, size=(n // 64, 64), dtype=np.uint8)
        out = ext.dequant_int4_packed_to_fp16(to_torch(p), 0.37, 128)
        assert np.array_equal(bits(out), bits(O.dequant_int4_packed_kernel_full(p, np.float32(0.37), 128)))
    with pytest.raises(RuntimeError):
        ext.dequant_int8_to_fp16(torch.zeros(4, dtype=torch.int8), 1.0)  # CPU tensor
    with pytest.raises(RuntimeError):
        ext.dequant_int8_to_fp16(torch.zeros(4, 4, dtype=torch.int8, device="cuda").t(), 1.0)  # non-contiguous


def test_cpu_tensors_fail_loudly(E):
    with pytest.raises(RuntimeError, match="MI355X"):
        E.quantize_int8_per_tensor(torch.randn(2, 2, 1, 8))
    with pytest.raises(RuntimeError, match="MI355X"):
        E.trim_kv_sliding_window(((torch.randn(1, 2, 9, 8), torch.randn(1, 2, 9, 8)),), 4)


def test_layer_append_and_growth(E):
    """Per-layer appends (the reference's own prefill loop, ops.py:339-342) through capacity
    growth, vs one fused prefill."""
    L, B, H, T, D = 2, 1, 4, 70, 64
    kv = to_torch(seeded_kv((L, 2, B, H, T, D), "f16", 9, "heavy"))
    a = E.QuantizedKVCache(L, "mixed")
    for l in range(L):
        for t in range(T):
            a.layers[l].append(kv[l, 0, :, :, t:t + 1], kv[l, 1, :, :, t:t + 1])
    b = E.QuantizedKVCache(L, "mixed")
    b.init_from_prompt_past(tuple((kv[l, 0], kv[l, 1]) for l in range(L)))
    for (ka, va), (kb, vb) in zip(a.to_past_key_values(), b.to_past_key_values()):
        assert torch.equal(ka, kb) and torch.equal(va, vb)
    assert a.estimated_bytes() == b.estimated_bytes() == O.estimated_bytes("mixed", L, B, H, T, D, 2)


# ------------------------------------------------------------------ eviction


@pytest.mark.parametrize("dtype", ["f16", "f32"])
@pytest.mark.parametrize("T,W", [(5, 8), (8, 8), (13, 8), (40, 1)])
def test_golden_sliding_window(E, g6, dtype, T, W):
    x = to_torch(g6[f"win.{dtype}.T{T}.W{W}.x"])
    v = x * 2
    (k2, v2), = E.trim_kv_sliding_window(((x, v),), W)
    assert np.array_equal(to_numpy(k2), g6[f"win.{dtype}.T{T}.W{W}.k"])
    assert np.array_equal(to_numpy(v2), g6[f"win.{dtype}.T{T}.W{W}.v"])
    if T <= W:
        assert k2 is x and v2 is v  # unchanged objects, as the reference


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,chunk,keep", [(40, 8, 8), (45, 8, 8), (6, 8, 8), (33, 4, 0), (300, 64, 16), (19, 64, 3)])
def test_golden_and_oracle_chunk_summary(E, g6, dtype, T, chunk, keep):
    key = f"chunk.{dtype}.T{T}.c{chunk}.k{keep}"
    x_np, ref = g6[key + ".x"], g6[key + ".k"]
    x = to_torch(x_np, dtype)
    (k2, v2), = E.chunk_summarize_kv(((x, -x),), chunk_size=chunk, keep_last=keep)
    orc = O.chunk_summarize_kv(x_np, chunk, keep, dtype=odt(dtype))
    assert np.array_equal(bits(k2), bits(orc))  # same summation order: bit-exact vs the oracle
    assert torch.equal(v2, -k2) or T <= keep
    # vs the reference's golden: recent tail exact, summaries within 1 storage ulp
    n_sum = ref.shape[-2] - min(keep, T) if T > keep else 0
    got = to_numpy(k2)
    assert np.array_equal(got[..., n_sum:, :], ref[..., n_sum:, :])
    a32 = O._widen(got[..., :n_sum, :], odt(dtype))
    b32 = O._widen(ref[..., :n_sum, :], odt(dtype))
    tol = {"f16": 2.0**-10, "bf16": 2.0**-7, "f32": 2.0**-22}[dtype]
    xmax = np.abs(O._widen(x_np, odt(dtype))).max()
    assert np.all(np.abs(a32 - b32) <= tol * np.abs(b32) + 8 * 2.0**-24 * xmax)


@pytest.mark.parametrize("shape,chunk,keep,W", [((6, 1, 8, 1000, 128), 64, 256, 256), ((4, 2, 3, 77, 5), 8, 5, 20),
                                                  ((2, 1, 4, 130, 24), 16, 2, 129)])
@pytest.mark.parametrize("dtype", DTYPES)
def test_oracle_eviction_multi_layer(E, shape, chunk, keep, W, dtype):
    """2L tensors per launch (pointer table), vector and generic kernels."""
    G, B, H, T, D = shape
    x_np = seeded_kv(shape, dtype, 31, "heavy")
    x = to_torch(x_np, dtype)
    past = tuple((x[2 * l], x[2 * l + 1]) for l in range(G // 2))
    res = E.chunk_summarize_kv(past, chunk, keep)
    orc = O.chunk_summarize_kv(x_np, chunk, keep, dtype=odt(dtype))
    for l in range(G // 2):
        assert np.array_equal(bits(res[l][0]), bits(orc[2 * l])) and np.array_equal(bits(res[l][1]), bits(orc[2 * l + 1]))
    res = E.trim_kv_sliding_window(past, W)
    for l in range(G // 2):
        assert np.array_equal(bits(res[l][0]), bits(x_np[2 * l][..., T - W:, :]))
        assert np.array_equal(bits(res[l][1]), bits(x_np[2 * l + 1][..., T - W:, :]))
    # strided inputs: windows of a larger cache (t not contiguous with the row)
    big = torch.zeros(G, B, H, T + 9, D, dtype=x.dtype, device="cuda")
    big[:, :, :, 4:T + 4] = x
    past = tuple((big[2 * l, :, :, 4:T + 4], big[2 * l + 1, :, :, 4:T + 4]) for l in range(G // 2))
    res = E.chunk_summarize_kv(past, chunk, keep)
    for l in range(G // 2):
        assert np.array_equal(bits(res[l][0]), bits(orc[2 * l]))


def test_incremental_staging_equals_full_dequant(E):
    """Scope row N1: persistent staging (dequantise only new tokens) returns the same bytes as a
    full re-dequantise, across appends and capacity growth; earlier views stay valid."""
    L, B, H, D = 3, 1, 4, 64
    kv = to_torch(seeded_kv((L, 2, B, H, 90, D), "f16", 23, "heavy"))
    inc = E.QuantizedKVCache(L, "mixed", incremental=True)
    ref = E.QuantizedKVCache(L, "mixed", incremental=False)
    first = None
    for T in (17, 18, 19, 40, 41, 90):  # growth past the initial capacity happens on the way
        t0 = len(inc.layers[0])
        chunk = tuple((kv[l, 0, :, :, t0:T], kv[l, 1, :, :, t0:T]) for l in range(L))
        for c in (inc, ref):
            c._k.append([k for k, _ in chunk])
            c._v.append([v for _, v in chunk])
        a, b = inc.to_past_key_values(), ref.to_past_key_values()
        if first is None:
            first = (a[0][0], a[0][0].clone())
        for (ka, va), (kb, vb) in zip(a, b):
            assert ka.shape == (B, H, T, D) and torch.equal(ka, kb) and torch.equal(va, vb)
        assert torch.equal(first[0], first[1])  # an old view is never rewritten
    assert inc._k.staged == 90 and inc.to_past_key_values()[1][1].data_ptr() == a[1][1].data_ptr()


@pytest.mark.parametrize("kind", ["int8", "int4"])
def test_quant_batch_strided_input(E, kind):
    """B > 1 with a batch stride that is not H * stride_h (every other batch row of a larger
    tensor): register-tile kernel is not eligible, swept-tile / generic paths must agree."""
    from efficient_llm_inference_amd import kernels
    for (G, B, H, T, D) in ((2, 3, 4, 9, 64), (1, 40, 8, 5, 128)):
        big_np = seeded_kv((G, 2 * B, H, T, D), "f16", 41, "heavy")
        x = to_torch(big_np)[:, ::2]
        x_np = big_np[:, ::2]
        q_ref, _, s32_ref = O.quantize_tokens(np.ascontiguousarray(x_np), kind)
        Dq = kernels.packed_dim(kind, D)
        store = torch.zeros(G, B, H, T, Dq, dtype=kernels.QDTYPE[kind], device="cuda")
        scales = torch.zeros(G, T, dtype=torch.float32, device="cuda")
        ws = torch.empty(G * T, dtype=torch.float32, device="cuda")
        kernels.quant_tokens(x, store, scales, ws, kind)
        assert np.array_equal(to_numpy(store), q_ref) and np.array_equal(bits(scales), bits(s32_ref))


def test_more_than_128_groups_chunked_launches(E):
    """Pointer tables hold 128 groups per launch: 130 separately allocated tensors exercise the
    second launch's base-pointer / scale-table / workspace offsets in every entry point."""
    from efficient_llm_inference_amd import kernels
    G, B, H, T, D = 130, 1, 2, 6, 64
    x_np = seeded_kv((G, B, H, T, D), "f16", 51, "heavy")
    xs = [to_torch(x_np[g]) for g in range(G)]  # separately allocated
    for kind in ("int8", "int4"):
        q_ref, _, s32_ref = O.quantize_tokens(x_np, kind)
        Dq = kernels.packed_dim(kind, D)
        for two_pass in (0, 1):
            store = torch.zeros(G, B, H, T, Dq, dtype=kernels.QDTYPE[kind], device="cuda")
            scales = torch.zeros(G, T, dtype=torch.float32, device="cuda")
            ws = torch.empty(G * T, dtype=torch.float32, device="cuda")
            E._lib.set_tunable("quant_force_two_pass", two_pass)
            try:
                kernels.quant_tokens(xs, store, scales, ws, kind)
            finally:
                E._lib.set_tunable("quant_force_two_pass", 0)
            assert np.array_equal(to_numpy(store), q_ref) and np.array_equal(bits(scales), bits(s32_ref)), (kind, two_pass)
        out = torch.empty(G, B, H, T, D, dtype=torch.float16, device="cuda")
        kernels.dequant_tokens(store, scales, out, kind)
        assert np.array_equal(bits(out), bits(O.dequantize_tokens(q_ref, s32_ref, kind, D, "f16")))
    past = tuple((xs[2 * l], xs[2 * l + 1]) for l in range(G // 2))
    res = E.chunk_summarize_kv(past, 2, 1)
    orc = O.chunk_summarize_kv(x_np, 2, 1)
    win = E.trim_kv_sliding_window(past, 4)
    from efficient_llm_inference_amd.cache import trim_kv_strided
    st = trim_kv_strided(past, 2, 2, 1)
    idx = O.keep_indices_strided(T, 2, 2, 1)
    for l in range(G // 2):
        for j in (0, 1):
            g = 2 * l + j
            assert np.array_equal(bits(res[l][j]), bits(orc[g]))
            assert np.array_equal(bits(win[l][j]), bits(x_np[g][..., T - 4:, :]))
            assert np.array_equal(bits(st[l][j]), bits(O.gather_tokens(x_np[g], idx)))


@pytest.mark.parametrize("cname", ["llamaish", "odd"])
@pytest.mark.parametrize("mode", ["int8", "int4", "mixed"])
def test_golden_cache_bf16_kv_fp16_compute(E, cname, mode):
    """bf16 KV tensors (Llama-family) through the container with fp16 compute_dtype."""
    from tests.conftest import load_golden
    g = load_golden("g5b_cache_bf16.npz")
    kv = to_torch(g[f"{cname}.kv"], "bf16")
    L, _, B, H, T1, D = kv.shape
    qc = E.QuantizedKVCache(n_layers=L, mode=mode, device="cuda", compute_dtype=torch.float16)
    qc.init_from_prompt_past(tuple((kv[l, 0, :, :, :T1 - 1], kv[l, 1, :, :, :T1 - 1]) for l in range(L)))
    qc.append_from_past(tuple((kv[l, 0], kv[l, 1]) for l in range(L)))
    past = qc.to_past_key_values()
    deq = torch.stack([torch.stack([k, v]) for k, v in past])
    assert deq.dtype == torch.float16 and np.array_equal(bits(deq), bits(g[f"{cname}.{mode}.deq"]))
    sc = torch.stack([torch.stack([torch.stack(l.k_scales), torch.stack(l.v_scales)]) for l in qc.layers])
    assert sc.dtype == torch.bfloat16 and np.array_equal(bits(sc), bits(g[f"{cname}.{mode}.scales"]))
    assert qc.estimated_bytes() == int(g[f"{cname}.{mode}.bytes"][0])


@pytest.mark.ab
@pytest.mark.parametrize("k", [2, 4, 8, 16])
def test_xcd_grouped_item_order_is_a_permutation(E, k):
    """dequant_xcd_group / quant_xcd_group only re-order which workgroup takes which chunk / tile
    (xcd_grouped_item: k consecutive items per XCD): results stay bit-exact for every k, with item counts
    that are and are not multiples of the 8 k window."""
    from efficient_llm_inference_amd import _lib, kernels
    for case in ((4, 1, 8, 300, 128), (2, 1, 8, 64, 128), (3, 1, 8, 1031, 128), (1, 2, 4, 37, 128)):
        G, B, H, T, D = case
        x_np = seeded_kv(case, "f16", seed=k * 131 + T, dist="heavy")
        for kind in ("int8", "int4"):
            q_ref, _, s32_ref = O.quantize_tokens(x_np, kind)
            _lib.set_tunable("quant_xcd_group", k)
            _lib.set_tunable("dequant_xcd_group", k)
            try:
                store, scales = _quant_via_kernels(E, x_np, "f16", kind, False, False, block=64, tile=0)
                assert np.array_equal(to_numpy(store[:, :, :, 1:T + 1]), q_ref) and np.array_equal(bits(scales[:, 1:T + 1]), bits(s32_ref))
                q = to_torch(q_ref)
                sc = to_torch(s32_ref)
                out = torch.empty(case, dtype=torch.float16, device="cuda")
                kernels.dequant_tokens(q, sc, out, kind)
                assert np.array_equal(bits(out), bits(O.dequantize_tokens(q_ref, s32_ref, kind, D, "f16"))), (case, kind, k)
            finally:
                _lib.set_tunable("quant_xcd_group", 0)
                _lib.set_tunable("dequant_xcd_group", 0)


@pytest.mark.parametrize("shape,chunk,keep", [((2, 2, 3, 1024, 128), 64, 0), ((2, 1, 8, 1000, 128), 64, 256), ((4, 1, 2, 333, 128), 16, 7),
                                              ((2, 2, 2, 520, 128), 32, 8), ((2, 1, 1, 256, 128), 64, 0), ((2, 1, 3, 255, 128), 64, 1)])
@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_chunk_pool_head_dim_128_bit_exact(E, shape, chunk, keep, dtype):
    """Chunk mean-pool at the Llama row shape (256-byte token rows): bit-exact with the oracle for chunk counts that
    are / are not multiples of 4, a ragged last chunk, keep_last 0, single-buffer and pointer-list inputs."""
    from efficient_llm_inference_amd import kernels
    G, B, H, T, D = shape
    x_np = seeded_kv(shape, dtype, seed=zlib.crc32(repr((shape, chunk, keep, dtype)).encode()), dist="heavy")
    x = to_torch(x_np, dtype)
    Tout = kernels.chunk_summary_len(T, chunk, keep)
    ref = O.chunk_summarize_kv(x_np, chunk, keep, dtype=odt(dtype))
    for src in (x, [x[g] for g in range(G)]):
        out = torch.empty(G, B, H, Tout, D, dtype=x.dtype, device="cuda")
        kernels.chunk_meanpool(src, out, chunk, keep)
        assert np.array_equal(bits(out), bits(ref))


@pytest.mark.parametrize("shape,chunk,keep", [
    ((2, 1, 8, 1000, 128), 64, 256),   # 4 lane groups x 16 rows (the default policy at the Llama row shape), ragged last chunk
    ((2, 2, 3, 520, 128), 32, 8),      # 4 x 8
    ((2, 1, 4, 700, 64), 64, 17),      # 8 x 8 (gpt2 head_dim)
    ((2, 1, 2, 900, 64), 128, 0),      # 8 x 16
    ((2, 1, 2, 300, 256), 32, 5),      # 2 x 16
    ((2, 2, 1, 130, 256), 16, 1),      # 2 x 8
    ((2, 1, 2, 64, 128), 64, 0),       # exactly one chunk
    ((2, 1, 2, 65, 128), 64, 0),       # one row in the last chunk
    ((2, 1, 2, 400, 128), 128, 3),     # 32 rows per group: not a wave shape, the per-lane-group kernel
])
@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_chunk_pool_one_wave_per_chunk_kernel(E, shape, chunk, keep, dtype):
    """pool_wave (default): one wave per output row, the chunk's rows split over lane groups, the sequential fp32 sum
    handed from group to group. Every (lane groups, rows per group) instantiation against the oracle and against the
    per-lane-group kernel (pool_wave = 0): equal bits; strided (windowed) inputs too."""
    from efficient_llm_inference_amd import _lib, kernels
    G, B, H, T, D = shape
    x_np = seeded_kv(shape, dtype, seed=zlib.crc32(repr((shape, chunk, keep, dtype, "wave")).encode()), dist="heavy")
    x = to_torch(x_np, dtype)
    Tout = kernels.chunk_summary_len(T, chunk, keep)
    ref = O.chunk_summarize_kv(x_np, chunk, keep, dtype=odt(dtype))
    big = torch.zeros(G, B, H, T + 11, D, dtype=x.dtype, device="cuda")
    big[:, :, :, 3:T + 3] = x
    outs = {}
    shipped = _lib.get_tunable("pool_wave")
    try:
        for wave in (1, 0):
            _lib.set_tunable("pool_wave", wave)
            for name, src in (("buffer", x), ("list", [x[g] for g in range(G)]), ("window", big[:, :, :, 3:T + 3])):
                out = torch.full((G, B, H, Tout, D), float("nan"), dtype=x.dtype, device="cuda")
                kernels.chunk_meanpool(src, out, chunk, keep)
                assert np.array_equal(bits(out), bits(ref)), (wave, name)
                outs[(wave, name)] = out
    finally:
        _lib.set_tunable("pool_wave", shipped)
    assert torch.equal(outs[(1, "buffer")].view(torch.int16), outs[(0, "buffer")].view(torch.int16))
