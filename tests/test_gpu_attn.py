"""Scope row N1 (second form) on the MI355X: one decode step of attention straight from the
INT8 / packed-INT4 store (kvq_decode_attn) vs the oracle's float64 restatement of the reference
pipeline (dequantise everything -> cat the new token -> softmax(q K^T) V).

Floating point, so a tolerance instead of bit equality: the kernel accumulates exact
integer x fp16 products in fp32 and skips the reference's rounding of the dequantised values to the
compute dtype (<= 2^-11 relative per element for fp16, 2^-8 for bf16), and the output is rounded to
the compute dtype once. TOL below = 2 output ulps + that slack.
"""
import zlib

import numpy as np
import pytest
import torch

from oracle import kvq_oracle as O
from tests.util import TD, to_numpy, to_torch

pytestmark = pytest.mark.gpu

TOL = {"f16": 2e-3, "bf16": 1.6e-2}


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available()
    from efficient_llm_inference_amd import _lib, kernels
    _lib.load()
    return kernels


def _as_f32(a, dtype):
    return O.bf16_bits_to_f32(a) if dtype == "bf16" else a.astype(np.float32)


def _rand(shape, dtype, rng, scale=1.0):
    x = (rng.standard_normal(shape) * scale).astype(np.float32)
    return O.f32_to_bf16_bits(x) if dtype == "bf16" else x.astype(np.float16)


def _run_case(K, B, Hq, Hkv, T, D, k_kind, v_kind, dtype, with_new, tcap_pad=3, q_scale=1.0, strided_q=False):
    rng = np.random.default_rng(zlib.crc32(repr((B, Hq, Hkv, T, D, k_kind, v_kind, dtype, with_new)).encode()))
    kv_np = [_rand((1, B, Hkv, T, D), dtype, rng), _rand((1, B, Hkv, T, D), dtype, rng)]
    # outlier channel in K, as real keys have
    if T:
        k32 = _as_f32(kv_np[0], dtype)
        k32[..., 3] *= 6.0
        kv_np[0] = O.f32_to_bf16_bits(k32) if dtype == "bf16" else k32.astype(np.float16)
    odt = "bf16" if dtype == "bf16" else None
    kq, _, ks = O.quantize_tokens(kv_np[0], k_kind, dtype=odt)
    vq, _, vs = O.quantize_tokens(kv_np[1], v_kind, dtype=odt)
    q_np = _rand((B, Hq, D), dtype, rng, q_scale)
    kn_np = _rand((B, Hkv, D), dtype, rng) if with_new else None
    vn_np = _rand((B, Hkv, D), dtype, rng) if with_new else None
    sm = 1.0 / np.sqrt(D)
    ref = O.decode_attention(_as_f32(q_np, dtype), kq[0], ks[0], k_kind, vq[0], vs[0], v_kind, D, sm,
                             None if kn_np is None else _as_f32(kn_np, dtype),
                             None if vn_np is None else _as_f32(vn_np, dtype), kv_dtype=dtype)

    Tcap = T + tcap_pad
    k_store = torch.zeros(B, Hkv, Tcap, kq.shape[-1], dtype=K.QDTYPE[k_kind], device="cuda")
    v_store = torch.zeros(B, Hkv, Tcap, vq.shape[-1], dtype=K.QDTYPE[v_kind], device="cuda")
    k_sc = torch.full((Tcap,), float("nan"), device="cuda")
    v_sc = torch.full((Tcap,), float("nan"), device="cuda")
    if T:
        k_store[:, :, :T] = to_torch(kq[0])
        v_store[:, :, :T] = to_torch(vq[0])
        k_sc[:T] = to_torch(ks[0])
        v_sc[:T] = to_torch(vs[0])
    # guard tokens past T hold garbage that must never be read into the result
    k_store[:, :, T:] = 77 if k_kind == "int8" else 0x7F
    v_store[:, :, T:] = 77 if v_kind == "int8" else 0x7F
    if strided_q:  # the layout HF's fused qkv projection hands over: [B, 1, 3E] split into q | k | v
        qkv = torch.zeros(B, 3 * Hq * D, dtype=TD[dtype], device="cuda")
        qkv[:, :Hq * D] = to_torch(q_np, dtype).reshape(B, Hq * D)
        q = qkv[:, :Hq * D].view(B, Hq, D)
    else:
        q = to_torch(q_np, dtype)
    kn = None if kn_np is None else to_torch(kn_np, dtype)
    vn = None if vn_np is None else to_torch(vn_np, dtype)
    out = torch.full((B, Hq, D), float("nan"), dtype=TD[dtype], device="cuda")
    ws = torch.empty(max(1, K.decode_attn_workspace(B, Hq, Hkv, T, D)), dtype=torch.float32, device="cuda")
    K.decode_attn(q, k_store, k_sc, k_kind, v_store, v_sc, v_kind, T, out, ws, sm, kn, vn)
    torch.cuda.synchronize()
    got = _as_f32(to_numpy(out), dtype).astype(np.float64)
    assert np.isfinite(got).all()
    err = np.abs(got - ref)
    bound = TOL[dtype] * (np.abs(ref) + np.abs(ref).max())
    assert (err <= bound).all(), (float(err.max()), float(np.abs(ref).max()))
    return float(err.max())


CASES = [  # B, Hq, Hkv, T, D
    (1, 12, 12, 300, 64),    # gpt2: one query head per kv head
    (1, 16, 16, 1024, 64),   # gpt2-medium at n_positions
    (1, 32, 8, 1000, 128),   # Llama-3-8B grouping (4 query heads per kv head), ragged split
    (2, 8, 4, 257, 128),     # batch 2, 2 query heads per kv head
    (1, 8, 1, 640, 128),     # multi-query: 8 query heads on one kv head
    (1, 6, 2, 130, 32),      # 3 per kv head (padded to 4), smallest head_dim
    (1, 4, 4, 200, 256),     # widest head_dim
    (3, 4, 2, 1, 64),        # a single stored token
    (1, 32, 8, 5000, 128),   # several splits of different fill
    (1, 16, 1, 300, 128),    # 16 query heads on one kv head: every MFMA column in use
    (2, 6, 2, 200, 128),     # 3 per kv head at head_dim 128: MFMA kernel with padded heads
    (1, 32, 8, 700, 64),     # Llama-3.2-1B grouping: 4 per kv head at head_dim 64 (MFMA kernel, 4 d per lane)
    (2, 16, 2, 131, 64),     # 8 per kv head at head_dim 64, ragged split
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("kinds", [("int8", "int8"), ("int8", "int4"), ("int4", "int4"), ("int4", "int8")])
def test_decode_attn_matches_oracle(K, case, kinds):
    for dtype, with_new in (("f16", True), ("f16", False), ("bf16", True)):
        _run_case(K, *case, kinds[0], kinds[1], dtype, with_new)


@pytest.mark.parametrize("case", [c for c in CASES if c[4] in (64, 128) and 3 <= c[1] // c[2] <= 8])
@pytest.mark.parametrize("kinds", [("int8", "int4"), ("int4", "int8")])
def test_decode_attn_valu_kernel_on_grouped_heads(K, case, kinds):
    """head_dim 128 with 3..16 query heads per kv head takes the MFMA kernel by default; the VALU
    kernel must give the same answer on those shapes (it serves every other head_dim)."""
    from efficient_llm_inference_amd import _lib
    _lib.set_tunable("attn_force_valu", 1)
    try:
        _run_case(K, *case, kinds[0], kinds[1], "f16", True)
    finally:
        _lib.set_tunable("attn_force_valu", 0)


def test_decode_attn_tiny_magnitudes(K):
    """V scales near the fp16 underflow range: P is normalised by the largest scale before the f16
    pack of the MFMA kernel, so small-magnitude caches keep their precision."""
    rng = np.random.default_rng(5)
    B, Hq, Hkv, T, D = 1, 8, 2, 300, 128
    k = rng.standard_normal((1, B, Hkv, T, D)).astype(np.float16)
    v = (rng.standard_normal((1, B, Hkv, T, D)) * 3e-4).astype(np.float16)
    kq, _, ks = O.quantize_tokens(k, "int8")
    vq, _, vs = O.quantize_tokens(v, "int4")
    q = rng.standard_normal((B, Hq, D)).astype(np.float16)
    sm = D ** -0.5
    ref = O.decode_attention(q, kq[0], ks[0], "int8", vq[0], vs[0], "int4", D, sm)
    out = torch.empty(B, Hq, D, dtype=torch.float16, device="cuda")
    ws = torch.empty(K.decode_attn_workspace(B, Hq, Hkv, T, D), dtype=torch.float32, device="cuda")
    K.decode_attn(to_torch(q), to_torch(kq[0]), to_torch(ks[0]), "int8", to_torch(vq[0]), to_torch(vs[0]), "int4", T, out, ws, sm)
    got = to_numpy(out).astype(np.float64)
    scale = np.abs(ref).max()
    assert scale > 0 and (np.abs(got - ref) <= 4e-3 * scale + 1e-7).all(), float(np.abs(got - ref).max() / scale)


@pytest.mark.parametrize("shape", [(1, 12, 12, 77, 64), (2, 32, 8, 300, 128)])
@pytest.mark.parametrize("kinds", [("int8", "int4"), ("int4", "int8")])
def test_decode_step_attends_then_appends(K, shape, kinds):
    """kvq_decode_step = kvq_decode_attn over the stored tokens + the new one, then the new token's
    K / V quantised into slot T: output within tolerance of the oracle, slot T bit-exact with the
    oracle's quantiser, every other slot untouched."""
    B, Hq, Hkv, T, D = shape
    rng = np.random.default_rng(T)
    k = rng.standard_normal((1, B, Hkv, T + 1, D)).astype(np.float16)
    v = rng.standard_normal((1, B, Hkv, T + 1, D)).astype(np.float16)
    kq, _, ks = O.quantize_tokens(k, kinds[0])
    vq, _, vs = O.quantize_tokens(v, kinds[1])
    q = rng.standard_normal((B, Hq, D)).astype(np.float16)
    sm = D ** -0.5
    ref = O.decode_attention(q, kq[0][:, :, :T], ks[0][:T], kinds[0], vq[0][:, :, :T], vs[0][:T], kinds[1], D, sm,
                             k[0][:, :, T].astype(np.float32), v[0][:, :, T].astype(np.float32))
    cap = T + 5
    k_store = torch.full((B, Hkv, cap, kq.shape[-1]), 9, dtype=K.QDTYPE[kinds[0]], device="cuda")
    v_store = torch.full((B, Hkv, cap, vq.shape[-1]), 9, dtype=K.QDTYPE[kinds[1]], device="cuda")
    k_sc = torch.full((cap,), -1.0, device="cuda")
    v_sc = torch.full((cap,), -1.0, device="cuda")
    k_store[:, :, :T] = to_torch(kq[0][:, :, :T])
    v_store[:, :, :T] = to_torch(vq[0][:, :, :T])
    k_sc[:T] = to_torch(ks[0][:T])
    v_sc[:T] = to_torch(vs[0][:T])
    qt = to_torch(q)
    kn, vn = to_torch(k[0][:, :, T].copy()), to_torch(v[0][:, :, T].copy())
    out = torch.empty_like(qt)
    ws = torch.empty(K.decode_attn_workspace(B, Hq, Hkv, cap, D), dtype=torch.float32, device="cuda")
    plan = K.DecodeStepPlan(qt, k_store, k_sc, kinds[0], v_store, v_sc, kinds[1], 1e-8)
    K.decode_step(plan, qt, kn, vn, T, out, ws, sm)
    torch.cuda.synchronize()
    got = to_numpy(out).astype(np.float64)
    assert (np.abs(got - ref) <= TOL["f16"] * (np.abs(ref) + np.abs(ref).max())).all()
    assert np.array_equal(to_numpy(k_store[:, :, :T + 1]), kq[0]) and np.array_equal(to_numpy(v_store[:, :, :T + 1]), vq[0])
    assert np.array_equal(to_numpy(k_sc[:T + 1]), ks[0]) and np.array_equal(to_numpy(v_sc[:T + 1]), vs[0])
    assert int((k_store[:, :, T + 1:] != 9).sum()) == 0 and int((v_store[:, :, T + 1:] != 9).sum()) == 0
    assert float((k_sc[T + 1:] + 1.0).abs().sum()) == 0.0 and float((v_sc[T + 1:] + 1.0).abs().sum()) == 0.0
    from efficient_llm_inference_amd._lib import KvqError
    with pytest.raises(KvqError):  # no capacity for the slot
        K.decode_step(plan, qt, kn, vn, cap, out, ws, sm)


def test_decode_attn_many_splits(K):
    """Contexts long enough that the merge handles more than one split per thread (MFMA kernel:
    313 splits of 128 tokens) and that the VALU kernel's splits hold several loop iterations."""
    _run_case(K, 1, 8, 2, 40000, 128, "int8", "int4", "f16", True)
    _run_case(K, 1, 2, 2, 40000, 64, "int4", "int8", "f16", True)
    _run_case(K, 4, 8, 8, 9000, 32, "int8", "int8", "f16", False)  # 4 x 8 x 9000 tokens: 2 iterations per split


@pytest.mark.parametrize("shape", [(1, 32, 8, 16384, 128), (2, 12, 12, 4000, 64), (8, 32, 8, 16384, 128)])
def test_decode_attn_full_size_token_permutation(K, shape):
    """Size-independent property at the benchmark shape (no oracle at this size): attention does not
    depend on the ORDER of the stored tokens, so permuting the rows of both stores together with
    their scales must reproduce the output up to fp32 summation order (different splits see different
    tokens). Also: appending a new token whose key equals a stored row changes the output continuously."""
    B, Hq, Hkv, T, D = shape
    g = torch.Generator(device="cuda").manual_seed(T)
    k_store = torch.randint(-127, 128, (B, Hkv, T, D), device="cuda", dtype=torch.int8, generator=g)
    v_store = torch.randint(0, 256, (B, Hkv, T, D // 2), device="cuda", dtype=torch.uint8, generator=g)
    k_sc = torch.rand(T, device="cuda", generator=g) * 0.02 + 0.002
    v_sc = torch.rand(T, device="cuda", generator=g) * 0.3 + 0.01
    q = torch.randn(B, Hq, D, device="cuda", dtype=torch.float16, generator=g)
    kn = torch.randn(B, Hkv, D, device="cuda", dtype=torch.float16, generator=g)
    vn = torch.randn(B, Hkv, D, device="cuda", dtype=torch.float16, generator=g)
    from efficient_llm_inference_amd import _lib
    need = K.decode_attn_workspace(B, Hq, Hkv, T, D)
    _lib.set_tunable("attn_stream_tpw", -1)  # the one-tile kernel writes more split partials than the streaming one
    need = max(need, K.decode_attn_workspace(B, Hq, Hkv, T, D))
    _lib.set_tunable("attn_stream_tpw", 0)
    ws = torch.empty(need, dtype=torch.float32, device="cuda")
    sm = D ** -0.5
    out1 = torch.empty_like(q)
    K.decode_attn(q, k_store, k_sc, "int8", v_store, v_sc, "int4", T, out1, ws, sm, kn, vn)
    perm = torch.randperm(T, device="cuda", generator=g)
    out2 = torch.empty_like(q)
    K.decode_attn(q, k_store[:, :, perm].contiguous(), k_sc[perm].contiguous(), "int8", v_store[:, :, perm].contiguous(),
                  v_sc[perm].contiguous(), "int4", T, out2, ws, sm, kn, vn)
    torch.cuda.synchronize()
    a, b = out1.float(), out2.float()
    assert torch.isfinite(a).all() and float((a - b).abs().max()) <= 2e-3 * float(a.abs().max())
    if B == 8:  # the batch-8 shape takes the streaming kernel by size: it must agree with the one-tile kernel
        _lib.set_tunable("attn_stream_tpw", -1)
        try:
            out4 = torch.empty_like(q)
            K.decode_attn(q, k_store, k_sc, "int8", v_store, v_sc, "int4", T, out4, ws, sm, kn, vn)
        finally:
            _lib.set_tunable("attn_stream_tpw", 0)
        torch.cuda.synchronize()
        assert float((out4.float() - a).abs().max()) <= 2e-3 * float(a.abs().max())
    # a prefix of the context + the rest folded in as ... the same tokens: T1 + (T - T1) split at an odd place
    out3 = torch.empty_like(q)
    K.decode_attn(q, k_store, k_sc, "int8", v_store, v_sc, "int4", T - 1, out3, ws, sm, kn, vn)
    assert float((out3.float() - a).abs().max()) <= 0.5 * float(a.abs().max())  # one token of T cannot move it far


def test_quantized_kv_cache_attend(K):
    """QuantizedKVCache.attend: the cache-level entry to the fused attention, with and without append."""
    import efficient_llm_inference_amd as E
    rng = np.random.default_rng(11)
    L, B, Hkv, Hq, T, D = 2, 1, 2, 8, 90, 128
    kv = rng.standard_normal((L, 2, B, Hkv, T + 1, D)).astype(np.float16)
    qc = E.QuantizedKVCache(L, "mixed")
    qc.reserve(T + 4)
    qc.init_from_prompt_past(tuple((to_torch(kv[l, 0][:, :, :T].copy()), to_torch(kv[l, 1][:, :, :T].copy())) for l in range(L)))
    q = rng.standard_normal((B, Hq, D)).astype(np.float16)
    for l in range(L):
        kq, _, ks = O.quantize_tokens(kv[l:l + 1, 0][:, :, :, :T], "int8")
        vq, _, vs = O.quantize_tokens(kv[l:l + 1, 1][:, :, :, :T], "int4")
        kn, vn = kv[l, 0][:, :, T], kv[l, 1][:, :, T]
        ref = O.decode_attention(q, kq[0], ks[0], "int8", vq[0], vs[0], "int4", D, D ** -0.5, kn.astype(np.float32), vn.astype(np.float32))
        got = qc.attend(l, to_torch(q), to_torch(kn.copy()), to_torch(vn.copy()), append=(l == 1))
        err = np.abs(to_numpy(got).astype(np.float64) - ref)
        assert (err <= TOL["f16"] * (np.abs(ref) + np.abs(ref).max())).all()
    assert len(qc.layers[0]) == T and len(qc.layers[1]) == T + 1
    kq1, _, _ = O.quantize_tokens(kv[1:2, 0], "int8")
    assert np.array_equal(to_numpy(qc._k.q[1, :, :, :T + 1]), kq1[0])  # the appended token is the oracle's


def test_decode_attn_only_new_token(K):
    # empty store: the softmax has the new token alone, out == v_new
    _run_case(K, 2, 8, 4, 0, 64, "int8", "int4", "f16", True)


def test_decode_attn_strided_query_and_large_logits(K):
    _run_case(K, 1, 12, 12, 700, 64, "int8", "int4", "f16", True, strided_q=True)
    # peaked softmax: logits of a few tens
    _run_case(K, 1, 8, 8, 900, 128, "int8", "int8", "f16", True, q_scale=8.0)


def test_decode_attn_rejects_bad_arguments(K):
    from efficient_llm_inference_amd._lib import KvqError
    q = torch.zeros(1, 4, 48, dtype=torch.float16, device="cuda")
    ks = torch.zeros(1, 4, 8, 48, dtype=torch.int8, device="cuda")
    sc = torch.ones(8, device="cuda")
    ws = torch.empty(4096, device="cuda")
    with pytest.raises(KvqError):  # head_dim 48 is not supported by the fused kernel
        K.decode_attn(q, ks, sc, "int8", ks, sc, "int8", 8, torch.empty_like(q), ws, 1.0)
    q = torch.zeros(1, 4, 64, dtype=torch.float32, device="cuda")
    ks = torch.zeros(1, 4, 8, 64, dtype=torch.int8, device="cuda")
    with pytest.raises(KvqError):  # fp32 queries are not supported
        K.decode_attn(q, ks, sc, "int8", ks, sc, "int8", 8, torch.empty_like(q), ws, 1.0)
    q = torch.zeros(1, 4, 64, dtype=torch.float16, device="cuda")
    with pytest.raises(KvqError):  # T beyond the store
        K.decode_attn(q, ks, sc, "int8", ks, sc, "int8", 9, torch.empty_like(q), ws, 1.0)
    with pytest.raises(KvqError):  # CPU tensors never reach the kernel
        K.decode_attn(q.cpu(), ks, sc, "int8", ks, sc, "int8", 8, torch.empty_like(q), ws, 1.0)
    with pytest.raises(KvqError):  # workspace too small
        K.decode_attn(q, ks, sc, "int8", ks, sc, "int8", 8, torch.empty_like(q), ws[:4], 1.0)


# ---------------------------------------------------------------------------- fused single launch

FUSED_SHAPES = [(128, 4), (128, 8), (64, 8), (32, 16)]  # (tokens per wave, waves per workgroup)


@pytest.fixture
def tunable():
    from efficient_llm_inference_amd import _lib
    touched = {}

    def set_(key, value):
        touched.setdefault(key, _lib.get_tunable(key))  # restore what the library shipped with, not 0
        _lib.set_tunable(key, value)
    yield set_
    for key, old in touched.items():
        _lib.set_tunable(key, old)


@pytest.mark.ab
@pytest.mark.parametrize("shape", FUSED_SHAPES)
@pytest.mark.parametrize("kinds", [("int8", "int4"), ("int4", "int8"), ("int8", "int8"), ("int4", "int4")])
def test_decode_attn_fused_shapes_match_oracle(K, tunable, shape, kinds):
    """Every (tokens per wave, waves per workgroup) shape of the fused single launch at head_dim 128:
    one workgroup, a ragged last workgroup, idle waves in the last workgroup, several batch rows."""
    tunable("attn_fused", 1)
    tunable("attn_fused_tc", shape[0])
    tunable("attn_fused_nw", shape[1])
    for case in [(1, 32, 8, 1000, 128), (2, 6, 2, 200, 128), (1, 16, 1, 300, 128), (1, 32, 8, 5000, 128), (3, 8, 2, 1, 128),
                 (1, 8, 2, 513, 128), (2, 32, 8, 2048, 128)]:
        for dtype, with_new in (("f16", True), ("f16", False), ("bf16", True)):
            _run_case(K, *case, kinds[0], kinds[1], dtype, with_new)


@pytest.mark.ab
@pytest.mark.parametrize("case", [c for c in CASES if c[4] in (64, 128) and 3 <= c[1] // c[2] <= 16])
def test_decode_attn_fused_default_shape_matches(K, tunable, case):
    """attn_fused = 1 with the shape picked by batch size (head_dim 64 included): same answers as the
    default partial + merge pair, which every other test of this file exercises."""
    tunable("attn_fused", 1)
    for kinds in (("int8", "int4"), ("int4", "int8")):
        _run_case(K, *case, kinds[0], kinds[1], "f16", True)


@pytest.mark.ab
@pytest.mark.parametrize("shape", FUSED_SHAPES)
def test_decode_attn_fused_workspace_reuse(K, tunable, shape):
    """The arrival words of the fused launch are never zeroed by anyone: a workspace full of garbage
    (all-ones, then whatever earlier calls left) must work, back-to-back launches on one stream with
    DIFFERENT inputs and context lengths must each merge their own partials (a stale partial or a
    miscounted ticket shows up as another call's output), and every (batch row, kv head) must be
    written exactly once per call."""
    tunable("attn_fused", 1)
    tunable("attn_fused_tc", shape[0])
    tunable("attn_fused_nw", shape[1])
    B, Hq, Hkv, D = 2, 8, 2, 128
    Tmax = 6000
    rng = np.random.default_rng(shape[0] + shape[1])
    sets = []
    for i in range(3):
        k = rng.standard_normal((1, B, Hkv, Tmax, D)).astype(np.float16)
        v = (rng.standard_normal((1, B, Hkv, Tmax, D)) * (1.0 + i)).astype(np.float16)
        kq, _, ks = O.quantize_tokens(k, "int8")
        vq, _, vs = O.quantize_tokens(v, "int4")
        q = rng.standard_normal((B, Hq, D)).astype(np.float16)
        sets.append(dict(kq=kq[0], ks=ks[0], vq=vq[0], vs=vs[0], q=q, kq_t=to_torch(kq[0]), ks_t=to_torch(ks[0]),
                         vq_t=to_torch(vq[0]), vs_t=to_torch(vs[0]), q_t=to_torch(q)))
    lengths = [6000, 700, 3000, 513, 5999, 1]
    sm = D ** -0.5
    ws = torch.empty(K.decode_attn_workspace(B, Hq, Hkv, Tmax, D), dtype=torch.float32, device="cuda")
    ws.view(torch.int32).fill_(-1)  # all-ones bit patterns in every arrival word and partial
    calls, outs = [], []
    for it in range(36):
        s, T = sets[it % 3], lengths[it % len(lengths)]
        out = torch.full((B, Hq, D), float("nan"), dtype=torch.float16, device="cuda")
        K.decode_attn(s["q_t"], s["kq_t"], s["ks_t"], "int8", s["vq_t"], s["vs_t"], "int4", T, out, ws, sm)
        calls.append((it % 3, T))
        outs.append(out)
    torch.cuda.synchronize()
    refs = {}
    for (si, T), out in zip(calls, outs):
        if (si, T) not in refs:
            s = sets[si]
            refs[(si, T)] = O.decode_attention(s["q"], s["kq"][:, :, :T], s["ks"][:T], "int8", s["vq"][:, :, :T], s["vs"][:T],
                                               "int4", D, sm)
        ref = refs[(si, T)]
        got = to_numpy(out).astype(np.float64)
        assert np.isfinite(got).all(), (si, T)
        assert (np.abs(got - ref) <= TOL["f16"] * (np.abs(ref) + np.abs(ref).max())).all(), (si, T)


@pytest.mark.parametrize("lds", [-1, pytest.param(0, marks=pytest.mark.ab)])  # 0 (A-B library): the register-staged kernel
@pytest.mark.parametrize("tc", [64, pytest.param(32, marks=pytest.mark.ab)])
@pytest.mark.parametrize("tpw", [pytest.param(1, marks=pytest.mark.ab), pytest.param(2, marks=pytest.mark.ab), pytest.param(3, marks=pytest.mark.ab), 5])  # shipped library: 1 / 3 / 9 tiles per wave are test_decode_attn_lds_staged_kernel's, 5 runs here (2 with the A-B suite)
def test_decode_attn_streaming_kernel_matches_oracle(K, tunable, tc, tpw, lds):
    """decode_attn_stream_mfma_k (one wave walks `tpw` tiles with the next tile's rows in flight, online
    softmax across tiles): forced on small shapes through the tunables — odd / even tile counts per wave, a
    ragged last tile, a last wave with fewer tiles, one-tile contexts, V scales that grow and shrink across
    tiles (the running reference scale), every kind pair, fp16 and bf16."""
    tunable("attn_stream_tpw", tpw)
    tunable("attn_lds", lds)
    if tc != 64:
        tunable("attn_stream_tc", tc)
        tunable("attn_lds", 0)  # 32-token tiles are the register-staged kernel's
    for case in [(1, 32, 8, 1000, 128), (2, 6, 2, 200, 128), (1, 16, 1, 300, 128), (1, 32, 8, 5000, 128), (3, 8, 2, 1, 128),
                 (1, 8, 2, 513, 128), (2, 32, 8, 2048, 128), (1, 8, 2, 64, 128), (1, 8, 2, 129, 128)]:
        # shipped library (the ring kernel serves both splits): every kind pair at 2 tiles per wave, the headline pair at 5
        every = tpw != 5 or lds == 0 or tc != 64
        for kinds in (("int8", "int4"), ("int4", "int8"), ("int8", "int8"), ("int4", "int4")) if every else (("int8", "int4"),):
            _run_case(K, *case, kinds[0], kinds[1], "f16", True)
        _run_case(K, *case, "int8", "int4", "bf16", True)
        _run_case(K, *case, "int8", "int4", "f16", False)
    # V magnitudes that differ by orders of magnitude between tiles
    rng = np.random.default_rng(tpw * 100 + tc)
    B, Hq, Hkv, T, D = 1, 8, 2, 700, 128
    k = rng.standard_normal((1, B, Hkv, T, D)).astype(np.float16)
    v = rng.standard_normal((1, B, Hkv, T, D)).astype(np.float32)
    v[:, :, :, 100:300] *= 1e-3
    v[:, :, :, 300:420] *= 30.0
    v[:, :, :, 420:] *= 0.05
    v = v.astype(np.float16)
    kq, _, ks = O.quantize_tokens(k, "int8")
    vq, _, vs = O.quantize_tokens(v, "int4")
    q = rng.standard_normal((B, Hq, D)).astype(np.float16)
    sm = D ** -0.5
    ref = O.decode_attention(q, kq[0], ks[0], "int8", vq[0], vs[0], "int4", D, sm)
    out = torch.empty(B, Hq, D, dtype=torch.float16, device="cuda")
    ws = torch.empty(K.decode_attn_workspace(B, Hq, Hkv, T, D), dtype=torch.float32, device="cuda")
    K.decode_attn(to_torch(q), to_torch(kq[0]), to_torch(ks[0]), "int8", to_torch(vq[0]), to_torch(vs[0]), "int4", T, out, ws, sm)
    got = to_numpy(out).astype(np.float64)
    assert (np.abs(got - ref) <= TOL["f16"] * (np.abs(ref) + np.abs(ref).max())).all()


LDS_CASES = [(1, 32, 8, 1000, 128), (2, 6, 2, 200, 128), (1, 16, 1, 300, 128), (1, 32, 8, 5000, 128), (3, 8, 2, 1, 128),
             (1, 8, 2, 513, 128), (2, 32, 8, 2048, 128), (1, 8, 2, 64, 128), (1, 8, 2, 129, 128), (8, 32, 8, 4100, 128)]


@pytest.mark.parametrize("which", [1, pytest.param(2, marks=pytest.mark.ab), pytest.param(3, marks=pytest.mark.ab)])  # 3: strided tile ownership
@pytest.mark.parametrize("tpw", [0, 1, 3, 9])
def test_decode_attn_lds_staged_kernel(K, tunable, tpw, which):
    """decode_attn_lds_mfma_k (a tile = whole 1 KiB LDS-DMA requests into a ring of LDS slots, operand fragments read
    back from the XOR-swizzled image): forced on small shapes through attn_stream_tpw — against the oracle for every
    kind pair, fp16 and bf16, with and without the new token, ragged last tiles, one-tile contexts, waves with fewer
    tiles than the ring is deep. tpw = 0: the split the library picks by size. which = 2 (A-B library):
    decode_attn_coal_mfma_k, the same whole-line requests into registers and one LDS image per wave."""
    from efficient_llm_inference_amd import _lib
    tunable("attn_stream_tpw", tpw)
    tunable("attn_lds", which)
    every = tpw == 3 or which != 1  # the shipped kernel: every kind pair at 3 tiles per wave, the headline pair (+ bf16) at the other splits
    for case in LDS_CASES:
        big = case[0] * case[3] > 20000  # the batch-8 case: the float64 oracle on the host is what takes the time
        if big and not ((every and which != 1) or tpw == 0):  # (shipped kernel: at the split the library picks; the A-B variants also at 3 tiles per wave)
            continue
        for kinds in (("int8", "int4"),) if big or not every else (("int8", "int4"), ("int4", "int8"), ("int8", "int8"), ("int4", "int4")):
            _run_case(K, *case, kinds[0], kinds[1], "f16", True)
        if not big:
            _run_case(K, *case, "int8", "int4", "bf16", True)
            if every:
                _run_case(K, *case, "int8", "int4", "f16", False)
    if tpw:
        _lib.kernel_log_clear()
        _run_case(K, 2, 32, 8, 1500, 128, "int8", "int4", "f16", True)
        # (4 query heads per kv head: the shipped kernel's one-score-output-per-tile instantiation, TG = 4)
        assert _lib.kernel_log()[0].startswith(("decode_attn_lds_mfma_k<8, 4, 64, true, 2, 4, 128, false>", "decode_attn_coal_mfma_k<", "decode_attn_lds_mfma_k<8, 4, 64, true, 2, 4, 128, false>")[which - 1]), _lib.kernel_log()
        _lib.kernel_log_clear()
        _run_case(K, 1, 16, 2, 700, 128, "int8", "int4", "f16", True)  # 8 query heads per kv head: one output per 16-token group
        if which != 2:
            assert _lib.kernel_log()[0].startswith("decode_attn_lds_mfma_k<8, 4, 64, true, 2, 1, 128, false>"), _lib.kernel_log()


@pytest.mark.ab
@pytest.mark.parametrize("tpw", [1, 3])
def test_decode_attn_lds_kernel_one_output_per_token_group(K, tunable, tpw):
    """A-B library, attn_tg = 1: the LDS-staged kernel with one score output per 16-token group at <= 4 query heads per
    kv head too (what shipped before the block-diagonal score product) — against the oracle, like the shipped path."""
    from efficient_llm_inference_amd import _lib
    tunable("attn_stream_tpw", tpw)
    tunable("attn_tg", 1)
    for case in LDS_CASES:
        _run_case(K, *case, "int8", "int4", "f16", True)
        if case[0] * case[3] <= 20000:
            _run_case(K, *case, "int8", "int8", "bf16", True)
    _lib.kernel_log_clear()
    _run_case(K, 2, 32, 8, 1500, 128, "int8", "int4", "f16", True)
    assert _lib.kernel_log()[0].startswith("decode_attn_lds_mfma_k<8, 4, 64, true, 2, 1, 128, false>"), _lib.kernel_log()


def test_streaming_plan_rejected_falls_back_to_one_tile_splits(K, tunable):
    """A forced one-tile-per-wave streaming plan needs T / 64 splits; beyond the merge kernels' 4096 the plan falls back
    to one-tile splits of 128 tokens — and the launcher must take the kernel of THAT layout (round 2's launcher kept
    the streaming kernel on the one-tile grid: tokens skipped or counted twice)."""
    from efficient_llm_inference_amd import _lib
    tunable("attn_stream_tpw", 1)
    B, Hq, Hkv, T, D = 1, 4, 1, 262144 + 64 * 3 + 5, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    ks = torch.randint(-127, 128, (B, Hkv, T, D), device="cuda", dtype=torch.int8, generator=g)
    vs = torch.randint(0, 256, (B, Hkv, T, D // 2), device="cuda", dtype=torch.uint8, generator=g)
    ksc = torch.rand(T, device="cuda", generator=g) * 0.02 + 0.002
    vsc = torch.rand(T, device="cuda", generator=g) * 0.3 + 0.01
    q = torch.randn(B, Hq, D, device="cuda", dtype=torch.float16, generator=g)
    out = torch.empty_like(q)
    ws = torch.empty(K.decode_attn_workspace(B, Hq, Hkv, T, D), dtype=torch.float32, device="cuda")
    _lib.kernel_log_clear()
    K.decode_attn(q, ks, ksc, "int8", vs, vsc, "int4", T, out, ws, D ** -0.5)
    assert _lib.kernel_log()[0].startswith("decode_attn_partial_mfma_k<8, 4, 128, 128"), _lib.kernel_log()
    # float64 attention over the dequantised values, on the device
    kf = ks[0, 0].double() * ksc.double()[:, None]
    v8 = torch.stack([(vs[0, 0] >> 4).to(torch.int16), (vs[0, 0] & 15).to(torch.int16)], dim=-1).reshape(T, D) - 8
    vf = v8.double() * vsc.double()[:, None]
    p = torch.softmax((q[0].double() @ kf.T) * D ** -0.5, dim=-1)
    ref = p @ vf
    err = (out[0].double() - ref).abs()
    assert bool((err <= 2e-3 * (ref.abs() + ref.abs().max())).all()), float(err.max())


@pytest.mark.ab
@pytest.mark.parametrize("which", [1, 2, 11, 13])  # 1: ring (shipped depth), 2: coalesced, 11 / 13: ring of depth 1 / 3
@pytest.mark.parametrize("tpw", [1, 2, 3, 5, 9])
def test_lds_staged_kernels_equal_the_register_staged_streaming_kernel(K, tunable, tpw, which):
    """A-B library: the LDS-DMA ring kernel and the coalesced kernel are BIT-identical to the register-staged streaming
    kernel (attn_lds = 0) on the same inputs: the fragments in registers are the same bytes, the arithmetic is the same
    code (AttnStream::consume)."""
    from efficient_llm_inference_amd import _lib
    tunable("attn_stream_tpw", tpw)
    tunable("attn_tg", 1)  # one score output per 16-token group on both sides (the shipped TG = 4 path sums l in another order)
    if which > 10:
        tunable("attn_lds_nb", which - 10)
        which = 1
    g = torch.Generator(device="cuda").manual_seed(11 + tpw)
    for (B, Hq, Hkv, T, D) in ((2, 32, 8, 1500, 128), (1, 16, 2, 449, 128)):
        for kk, vk in (("int8", "int4"), ("int4", "int8"), ("int8", "int8")):
            ks = torch.randint(-127, 128, (B, Hkv, T + 5, K.packed_dim(kk, D)), device="cuda", dtype=torch.int8, generator=g).view(K.QDTYPE[kk])
            vs = torch.randint(-127, 128, (B, Hkv, T + 5, K.packed_dim(vk, D)), device="cuda", dtype=torch.int8, generator=g).view(K.QDTYPE[vk])
            ksc = torch.rand(T + 5, device="cuda", generator=g) * 0.02 + 0.002
            vsc = torch.rand(T + 5, device="cuda", generator=g) * 0.3 + 0.01
            q = torch.randn(B, Hq, D, device="cuda", dtype=torch.float16, generator=g)
            kn = torch.randn(B, Hkv, D, device="cuda", dtype=torch.float16, generator=g)
            vn = torch.randn(B, Hkv, D, device="cuda", dtype=torch.float16, generator=g)
            outs, names = [], []
            for lds in (which, 0):
                tunable("attn_lds", lds)
                ws = torch.full((K.decode_attn_workspace(B, Hq, Hkv, T, D),), float("nan"), dtype=torch.float32, device="cuda")
                out = torch.empty(B, Hq, D, dtype=torch.float16, device="cuda")
                _lib.kernel_log_clear()
                K.decode_attn(q, ks, ksc, kk, vs, vsc, vk, T, out, ws, D ** -0.5, kn, vn)
                names.append(_lib.kernel_log()[0])
                outs.append(out)
            assert names[0].startswith(("decode_attn_lds_mfma_k<", "decode_attn_coal_mfma_k<")[which - 1]) and names[1].startswith("decode_attn_stream_mfma_k<"), names
            assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), (B, Hq, Hkv, T, kk, vk, tpw)


@pytest.mark.parametrize("fused", [0, pytest.param(1, marks=pytest.mark.ab)])
@pytest.mark.parametrize("append", [False, True])
def test_decode_step_layers_equals_per_layer_calls(K, tunable, append, fused):
    """kvq_decode_step_layers: L launches behind one host call == L separate kvq_decode_attn /
    kvq_decode_step calls, bit for bit (same kernels, same workspace), incl. the appended slot."""
    if fused:
        tunable("attn_fused", fused)
    L, B, Hq, Hkv, T, D = 3, 2, 16, 4, 777, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    cap = T + 2

    def stores():
        ks = torch.randint(-127, 128, (L, B, Hkv, cap, D), device="cuda", dtype=torch.int8, generator=torch.Generator(device="cuda").manual_seed(1))
        vs = torch.randint(0, 256, (L, B, Hkv, cap, D // 2), device="cuda", dtype=torch.uint8, generator=torch.Generator(device="cuda").manual_seed(2))
        ksc = torch.rand(L, cap, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 0.02 + 0.002
        vsc = torch.rand(L, cap, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4)) * 0.3 + 0.01
        return ks, vs, ksc, vsc
    q = torch.randn(L, B, Hq, D, device="cuda", dtype=torch.float16, generator=g)
    kn = torch.randn(L, B, Hkv, D, device="cuda", dtype=torch.float16, generator=g)
    vn = torch.randn(L, B, Hkv, D, device="cuda", dtype=torch.float16, generator=g)
    ws = torch.empty(K.decode_attn_workspace_cap(B, Hq, Hkv, cap, D), dtype=torch.float32, device="cuda")
    sm = D ** -0.5
    k1, v1, ks1, vs1 = stores()
    out1 = torch.empty_like(q)
    for i in range(L):
        if append:
            plan = K.DecodeStepPlan(q[i], k1[i], ks1[i], "int8", v1[i], vs1[i], "int4", 1e-8)
            K.decode_step(plan, q[i], kn[i], vn[i], T, out1[i], ws, sm)
        else:
            K.decode_attn(q[i], k1[i], ks1[i], "int8", v1[i], vs1[i], "int4", T, out1[i], ws, sm, kn[i], vn[i])
    k2, v2, ks2, vs2 = stores()
    out2 = torch.empty_like(q)
    plan = K.DecodeLayersPlan(q, kn, vn, out2, k2, ks2, "int8", v2, vs2, "int4")
    K.decode_step_layers(plan, T, ws, sm, append=append)
    torch.cuda.synchronize()
    assert torch.equal(out1.view(torch.int16), out2.view(torch.int16))
    assert torch.equal(k1, k2) and torch.equal(v1, v2) and torch.equal(ks1, ks2) and torch.equal(vs1, vs2)
    if append:  # slot T now holds the quantised new token, bit-exact with the oracle's quantiser
        kq, _, ksc = O.quantize_tokens(to_numpy(kn)[:, :, :, None, :], "int8")
        assert np.array_equal(to_numpy(k2[:, :, :, T]), kq[:, :, :, 0]) and np.array_equal(to_numpy(ks2[:, T]), ksc[:, 0])
    from efficient_llm_inference_amd._lib import KvqError
    with pytest.raises(KvqError):
        K.decode_step_layers(plan, cap + 1, ws, sm)


@pytest.mark.parametrize("shape", [(1, 12, 12, 64), (2, 32, 8, 128), (1, 8, 2, 128)])
@pytest.mark.parametrize("kinds", [("int8", "int8"), ("int8", "int4"), ("int4", "int8")])
def test_decode_step_dev_reads_token_count_from_device(K, shape, kinds):
    """kvq_decode_step_dev: ONE set of launch arguments (upper bound as the host's T) serves every context length;
    the stored-token count is read from a device word. For a sweep of counts written into that word: output within
    tolerance of the oracle, slot T quantised bit-exactly, nothing else touched — VALU and MFMA kernels."""
    B, Hq, Hkv, D = shape
    cap, bound = 420, 400
    rng = np.random.default_rng(Hq * 7 + D)
    k = rng.standard_normal((1, B, Hkv, cap, D)).astype(np.float16)
    v = rng.standard_normal((1, B, Hkv, cap, D)).astype(np.float16)
    kq, _, ks = O.quantize_tokens(k, kinds[0])
    vq, _, vs = O.quantize_tokens(v, kinds[1])
    q = rng.standard_normal((B, Hq, D)).astype(np.float16)
    sm = D ** -0.5
    qt = to_torch(q)
    ws = torch.empty(K.decode_attn_workspace_cap(B, Hq, Hkv, cap, D), dtype=torch.float32, device="cuda")
    t_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
    for T in (1, 2, 77, 128, 129, 256, 300, 400):
        k_store = torch.full((B, Hkv, cap, kq.shape[-1]), 9, dtype=K.QDTYPE[kinds[0]], device="cuda")
        v_store = torch.full((B, Hkv, cap, vq.shape[-1]), 9, dtype=K.QDTYPE[kinds[1]], device="cuda")
        k_sc = torch.full((cap,), -1.0, device="cuda")
        v_sc = torch.full((cap,), -1.0, device="cuda")
        k_store[:, :, :T] = to_torch(kq[0][:, :, :T])
        v_store[:, :, :T] = to_torch(vq[0][:, :, :T])
        k_sc[:T] = to_torch(ks[0][:T])
        v_sc[:T] = to_torch(vs[0][:T])
        kn, vn = to_torch(k[0][:, :, T].copy()), to_torch(v[0][:, :, T].copy())
        out = torch.full_like(qt, float("nan"))
        plan = K.DecodeStepPlan(qt, k_store, k_sc, kinds[0], v_store, v_sc, kinds[1], 1e-8)
        t_dev.fill_(T)
        K.decode_step_dev(plan, qt, kn, vn, t_dev, bound, out, ws, sm)
        torch.cuda.synchronize()
        ref = O.decode_attention(q, kq[0][:, :, :T], ks[0][:T], kinds[0], vq[0][:, :, :T], vs[0][:T], kinds[1], D, sm,
                                 k[0][:, :, T].astype(np.float32), v[0][:, :, T].astype(np.float32))
        got = to_numpy(out).astype(np.float64)
        assert np.isfinite(got).all() and (np.abs(got - ref) <= TOL["f16"] * (np.abs(ref) + np.abs(ref).max())).all(), T
        assert np.array_equal(to_numpy(k_store[:, :, :T + 1]), kq[0][:, :, :T + 1]) and np.array_equal(to_numpy(v_store[:, :, :T + 1]), vq[0][:, :, :T + 1]), T
        assert np.array_equal(to_numpy(k_sc[:T + 1]), ks[0][:T + 1]) and np.array_equal(to_numpy(v_sc[:T + 1]), vs[0][:T + 1])
        assert int((k_store[:, :, T + 1:] != 9).sum()) == 0 and float((k_sc[T + 1:] + 1.0).abs().sum()) == 0.0
    from efficient_llm_inference_amd._lib import KvqError
    with pytest.raises(KvqError):
        K.decode_step_dev(plan, qt, kn, vn, t_dev, cap, out, ws, sm)  # bound outside the store
    with pytest.raises(KvqError):
        K.decode_step_dev(plan, qt, kn, vn, t_dev.long(), bound, out, ws, sm)  # not int32


@pytest.mark.ab
@pytest.mark.parametrize("stream", [(-1, 64), (2, 64), (3, 32)])
def test_decode_attn_int8_keys_through_int8_mfma(K, tunable, stream):
    """attn_k_i8: INT8 keys go into v_mfma_i32_16x16x64_i8 as stored (no byte -> f16 conversion), the query as two
    int8 planes. One-tile and streaming kernels, INT8 and INT4 values, fp16 and bf16, padded heads, peaked softmax."""
    tunable("attn_k_i8", 1)
    tunable("attn_lds", 0)  # the register-staged streaming kernel
    tunable("attn_stream_tpw", stream[0])
    tunable("attn_stream_tc", stream[1])
    for case in [(1, 32, 8, 1000, 128), (2, 6, 2, 200, 128), (1, 16, 1, 300, 128), (1, 32, 8, 5000, 128), (3, 8, 2, 1, 128),
                 (1, 8, 2, 513, 128), (2, 32, 8, 2048, 128)]:
        for vk in ("int4", "int8"):
            _run_case(K, *case, "int8", vk, "f16", True)
        _run_case(K, *case, "int8", "int4", "bf16", True)
        _run_case(K, *case, "int8", "int4", "f16", False)
    _run_case(K, 1, 8, 2, 900, 128, "int8", "int8", "f16", True, q_scale=8.0)
    _run_case(K, 1, 8, 2, 900, 128, "int8", "int4", "f16", True, q_scale=1e-3)


@pytest.mark.ab
@pytest.mark.parametrize("case", [(1, 32, 8, 16384, 128), (8, 32, 8, 4100, 128), (1, 12, 12, 300, 64), (1, 4, 4, 200, 256),
                                  (2, 6, 2, 130, 32), (3, 4, 2, 1, 64), (1, 32, 8, 32768, 128), (1, 32, 8, 700, 64),
                                  (16, 8, 8, 300, 128)])  # last: a new token too large for the register-resident quantise
def test_merge_one_round_trip_equals_chained_merge(K, tunable, case):
    """attn_merge_fast (default): the merge kernel that requests all its operands up front computes the same
    arithmetic in the same order as the chained one — equal output BITS, with and without a new token, fp16 and
    bf16, and with a device-side token count whose dead splits hold NaN in the workspace."""
    B, Hq, Hkv, T, D = case
    g = torch.Generator(device="cuda").manual_seed(T + D)
    for dtype in (torch.float16, torch.bfloat16):
        k_store = torch.randint(-127, 128, (B, Hkv, T + 130, D), dtype=torch.int8, device="cuda", generator=g)
        v_store = torch.randint(0, 256, (B, Hkv, T + 130, D // 2), dtype=torch.uint8, device="cuda", generator=g)
        k_sc = torch.rand(T + 130, device="cuda", generator=g) * 0.02 + 1e-3
        v_sc = torch.rand(T + 130, device="cuda", generator=g) * 0.2 + 1e-3
        q = torch.randn(B, Hq, D, device="cuda", generator=g).to(dtype)
        kn = torch.randn(B, Hkv, D, device="cuda", generator=g).to(dtype)
        vn = torch.randn(B, Hkv, D, device="cuda", generator=g).to(dtype)
        ws = torch.empty(K.decode_attn_workspace_cap(B, Hq, Hkv, T + 130, D), dtype=torch.float32, device="cuda")
        for with_new in (True, False):
            outs = []
            for fast in (1, 0):
                tunable("attn_merge_fast", fast)
                out = torch.full((B, Hq, D), float("nan"), dtype=dtype, device="cuda")
                ws.fill_(float("nan"))
                K.decode_attn(q, k_store, k_sc, "int8", v_store, v_sc, "int4", T, out, ws, D ** -0.5,
                              kn if with_new else None, vn if with_new else None)
                torch.cuda.synchronize()
                assert torch.isfinite(out.float()).all()
                outs.append(out)
            assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), (dtype, with_new)
        if dtype == torch.float16:  # device-side count below the host's bound: the splits past it are dead
            t_dev = torch.full((1,), T, dtype=torch.int32, device="cuda")
            outs = []
            for fast in (1, 0):
                tunable("attn_merge_fast", fast)
                ks, vs = k_store.clone(), v_store.clone()
                ksc, vsc = k_sc.clone(), v_sc.clone()
                plan = K.DecodeStepPlan(q, ks, ksc, "int8", vs, vsc, "int4", 1e-8)
                out = torch.full((B, Hq, D), float("nan"), dtype=dtype, device="cuda")
                ws.fill_(float("nan"))
                K.decode_step_dev(plan, q, kn, vn, t_dev, T + 129, out, ws, D ** -0.5)
                torch.cuda.synchronize()
                assert torch.isfinite(out.float()).all()
                outs.append((out, ks, vs, ksc, vsc))
            for a, b in zip(outs[0], outs[1]):  # output, both stores and both scale tables (slot T written by the step)
                assert torch.equal(a, b)


@pytest.mark.parametrize("case", [(2, 32, 8, 1500, 3), (1, 8, 2, 1000, 1), (3, 6, 2, 449, 2), (1, 4, 1, 64, 1), (5, 32, 8, 1024, 1),
                                  (2, 32, 8, 1500, 3, 64), (1, 12, 12, 1000, 1, 64), (3, 6, 2, 449, 2, 64)])  # (B, Hq, Hkv, T, tiles per wave[, head_dim])
def test_merge_by_one_wave_equals_the_workgroup_merge(K, tunable, case):
    """attn_merge_wave (default 1): up to 16 split partials per head at head_dim 128 — what the LDS-staged kernel leaves — are
    merged by ONE WAVE per head (weights by v_readlane, no LDS table, no barrier); 0 = the 256-thread workgroup per head.
    Same operands in the same order: equal output BITS — with and without the new token, fp16 and bf16, through
    decode_attn and through decode_step (whose merge launch also quantise-appends the new token: equal stores and scales)."""
    B, Hq, Hkv, T, tpw = case[:5]
    D = case[5] if len(case) > 5 else 128  # 64 (round 4): the head_dim-64 ring kernel's partials, merge_one_wave<64>
    from efficient_llm_inference_amd import _lib
    tunable("attn_stream_tpw", tpw)
    g = torch.Generator(device="cuda").manual_seed(T + tpw)
    for dtype in (torch.float16, torch.bfloat16):
        k_store = torch.randint(-127, 128, (B, Hkv, T + 4, D), dtype=torch.int8, device="cuda", generator=g)
        v_store = torch.randint(0, 256, (B, Hkv, T + 4, D // 2), dtype=torch.uint8, device="cuda", generator=g)
        k_sc = torch.rand(T + 4, device="cuda", generator=g) * 0.02 + 1e-3
        v_sc = torch.rand(T + 4, device="cuda", generator=g) * 0.2 + 1e-3
        q = torch.randn(B, Hq, D, device="cuda", generator=g).to(dtype)
        kn = torch.randn(B, Hkv, D, device="cuda", generator=g).to(dtype)
        vn = torch.randn(B, Hkv, D, device="cuda", generator=g).to(dtype)
        ws = torch.empty(K.decode_attn_workspace_cap(B, Hq, Hkv, T + 4, D), dtype=torch.float32, device="cuda")
        for with_new in (True, False):
            outs = []
            for wave in (1, 0):
                tunable("attn_merge_wave", wave)
                out = torch.full((B, Hq, D), float("nan"), dtype=dtype, device="cuda")
                ws.fill_(float("nan"))
                _lib.kernel_log_clear()
                K.decode_attn(q, k_store, k_sc, "int8", v_store, v_sc, "int4", T, out, ws, D ** -0.5,
                              kn if with_new else None, vn if with_new else None)
                torch.cuda.synchronize()
                log = _lib.kernel_log()
                assert log[0].startswith("decode_attn_lds_mfma_k<") and log[1] == "decode_attn_merge_fast_k<2, false>", log
                assert torch.isfinite(out.float()).all()
                outs.append(out)
            assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), (dtype, with_new)
        res = []
        for wave in (1, 0):  # the whole step: attention + the new token quantised into slot T by the merge launch's last workgroups
            tunable("attn_merge_wave", wave)
            ks, vs, ksc, vsc = k_store.clone(), v_store.clone(), k_sc.clone(), v_sc.clone()
            plan = K.DecodeStepPlan(q, ks, ksc, "int8", vs, vsc, "int4", 1e-8)
            out = torch.full((B, Hq, D), float("nan"), dtype=dtype, device="cuda")
            K.decode_step(plan, q, kn, vn, T, out, ws, D ** -0.5)
            torch.cuda.synchronize()
            res.append((out, ks, vs, ksc, vsc))
        for a_, b_ in zip(res[0], res[1]):
            assert torch.equal(a_, b_)
        assert not torch.equal(res[0][1][:, :, T], k_store[:, :, T])  # slot T was written


@pytest.mark.parametrize("kinds", [("int8", "int4"), ("int8", "int8"), ("int4", "int4")])
@pytest.mark.parametrize("shape", [(1, 32, 8, 16384, 128), (8, 32, 8, 16384, 128), (1, 16, 16, 4096, 64)])
def test_decode_attn_full_size_against_float64_attention_on_the_device(K, shape, kinds):
    """The benchmark shapes against an independent computation (the CPU oracle does not finish at this size): the
    stores dequantised to fp16 by the token-table kernel (bit-exact with the reference, test_gpu_fullsize), then plain
    float64 softmax(q K^T) V with torch ops on the GPU — the oracle's definition of the attention, at
    Llama-3-8B seq 16K (batch 1 and 8) and gpt2-medium seq 4K. Peaked and flat softmax rows both occur (K scales
    spread over a decade). Same tolerance as the small-shape oracle tests."""
    B, Hq, Hkv, T, D = shape
    g = torch.Generator(device="cuda").manual_seed(T + D + B)

    def store(kind):
        if kind == "int8":
            return torch.randint(-127, 128, (B, Hkv, T, D), dtype=torch.int8, device="cuda", generator=g)
        return torch.randint(0, 256, (B, Hkv, T, D // 2), dtype=torch.uint8, device="cuda", generator=g)

    k_store, v_store = store(kinds[0]), store(kinds[1])
    qmax = {"int8": 127.0, "int4": 7.0}
    k_sc = (10.0 ** (torch.rand(T, device="cuda", generator=g) - 1.0)) * (4.0 / qmax[kinds[0]])
    v_sc = (torch.rand(T, device="cuda", generator=g) * 0.9 + 0.1) * (3.0 / qmax[kinds[1]])
    k_sc, v_sc = k_sc.half().float(), v_sc.half().float()  # stored scales are fp16 values (reference: scale.to(x.dtype))
    q = torch.randn(B, Hq, D, device="cuda", generator=g).half()
    kn = torch.randn(B, Hkv, D, device="cuda", generator=g).half()
    vn = torch.randn(B, Hkv, D, device="cuda", generator=g).half()
    sm = D ** -0.5
    out = torch.full((B, Hq, D), float("nan"), dtype=torch.float16, device="cuda")
    ws = torch.empty(K.decode_attn_workspace(B, Hq, Hkv, T, D), dtype=torch.float32, device="cuda")
    K.decode_attn(q, k_store, k_sc, kinds[0], v_store, v_sc, kinds[1], T, out, ws, sm, kn, vn)

    kd = torch.empty(1, B, Hkv, T, D, dtype=torch.float16, device="cuda")
    vd = torch.empty(1, B, Hkv, T, D, dtype=torch.float16, device="cuda")
    K.dequant_tokens(k_store.unsqueeze(0), k_sc.unsqueeze(0), kd, kinds[0])
    K.dequant_tokens(v_store.unsqueeze(0), v_sc.unsqueeze(0), vd, kinds[1])
    nq = Hq // Hkv
    ref = torch.empty(B, Hq, D, dtype=torch.float64, device="cuda")
    for b in range(B):  # one batch row at a time: float64 copies of one row's K / V only
        kf = torch.cat([kd[0, b], kn[b].unsqueeze(1)], dim=1).double()  # [Hkv, T + 1, D]
        vf = torch.cat([vd[0, b], vn[b].unsqueeze(1)], dim=1).double()
        qf = q[b].double().view(Hkv, nq, D)
        s = torch.einsum("hqd,htd->hqt", qf, kf) * sm
        p = torch.softmax(s, dim=-1)
        ref[b] = torch.einsum("hqt,htd->hqd", p, vf).reshape(Hq, D)
        del kf, vf, s, p
    torch.cuda.synchronize()
    got = out.double()
    assert torch.isfinite(got).all()
    bound = TOL["f16"] * (ref.abs() + ref.abs().max())
    assert bool(((got - ref).abs() <= bound).all()), float(((got - ref).abs() / bound).max())


@pytest.mark.ab
@pytest.mark.parametrize("tpw", [1, 2, 3, 8])
@pytest.mark.parametrize("ki8", [0, 1])
def test_streaming_kernel_rolling_requests_equal_whole_tile_requests(K, tunable, tpw, ki8):
    """attn_stream_roll (default): a tile's registers are re-requested piece by piece for the tile after next while
    the tile is being reduced. Only the load schedule changes: output bits equal the whole-tile schedule's — odd and
    even tile counts, a ragged last tile, waves with one tile, every kind pair, both key paths."""
    tunable("attn_stream_tpw", tpw)
    tunable("attn_lds", 0)  # the register-staged streaming kernel
    tunable("attn_stream_tc", 64)
    tunable("attn_k_i8", ki8)
    for B, Hq, Hkv, T, D in [(1, 32, 8, 1000, 128), (2, 6, 2, 200, 128), (8, 32, 8, 4100, 128), (1, 8, 2, 64, 128), (3, 8, 2, 1, 128)]:
        g = torch.Generator(device="cuda").manual_seed(T + tpw)
        for kinds in (("int8", "int4"), ("int4", "int8"), ("int8", "int8"), ("int4", "int4")):
            def store(kind):
                if kind == "int8":
                    return torch.randint(-127, 128, (B, Hkv, T + 3, D), dtype=torch.int8, device="cuda", generator=g)
                return torch.randint(0, 256, (B, Hkv, T + 3, D // 2), dtype=torch.uint8, device="cuda", generator=g)
            k_store, v_store = store(kinds[0]), store(kinds[1])
            k_sc = torch.rand(T + 3, device="cuda", generator=g) * 0.02 + 1e-3
            v_sc = torch.rand(T + 3, device="cuda", generator=g) * 0.2 + 1e-3
            k_sc[T:] = float("nan")  # guard rows past T are never read into the result
            v_sc[T:] = float("nan")
            q = torch.randn(B, Hq, D, device="cuda", generator=g).half()
            ws = torch.empty(K.decode_attn_workspace(B, Hq, Hkv, T, D), dtype=torch.float32, device="cuda")
            outs = []
            for roll in (1, 0):
                tunable("attn_stream_roll", roll)
                out = torch.full((B, Hq, D), float("nan"), dtype=torch.float16, device="cuda")
                K.decode_attn(q, k_store, k_sc, kinds[0], v_store, v_sc, kinds[1], T, out, ws, D ** -0.5, None, None)
                torch.cuda.synchronize()
                assert torch.isfinite(out.float()).all()
                outs.append(out)
            assert torch.equal(outs[0], outs[1]), (B, Hq, Hkv, T, D, kinds)


@pytest.mark.ab
@pytest.mark.parametrize("case", [(2, 32, 8, 1500, 3), (1, 8, 2, 1000, 1), (3, 6, 2, 449, 2), (1, 4, 1, 64, 1), (5, 32, 8, 1024, 1),
                                  (8, 32, 8, 16384, 0), (2, 6, 2, 900, 2)])
def test_merge_inside_the_partial_launch_equals_the_two_launch_path(K, tunable, case):
    """attn_fold (round 4, A-B library: measured slower, profiles/r04b_*): the LDS-staged kernel's waves store their partials
    write-through, take a ticket per (batch row, kv head), and the wave that draws the last ticket merges the head group itself
    (merge_group_one_wave) — ONE launch per layer. attn_fold = 1: in kvq_decode_step_layers (one memset of the arrival words
    per host call), 2: in kvq_decode_attn too (a memset per call). Same operands in the same order as the merge kernels: equal output BITS with attn_fold = 0
    (two launches) — with and without the new token, fp16 and bf16, every kind pair, a workspace full of NaNs (the arrival words
    need no initialisation by the caller), the same workspace reused by 5 layers and by repeated calls (the words end a launch at
    zero), and more query heads than the merge takes (falls back to two launches)."""
    B, Hq, Hkv, T, tpw = case
    D, L = 128, 5
    from efficient_llm_inference_amd import _lib
    tunable("attn_stream_tpw", tpw)
    g = torch.Generator(device="cuda").manual_seed(T + tpw)
    for dtype, kinds in ((torch.float16, ("int8", "int4")), (torch.bfloat16, ("int8", "int4")), (torch.float16, ("int4", "int8")),
                         (torch.float16, ("int8", "int8")), (torch.float16, ("int4", "int4"))):
        if B * T > 50000 and (dtype != torch.float16 or kinds != ("int8", "int4")):
            continue
        kd, vd = (D if k == "int8" else D // 2 for k in kinds)
        ks_ = torch.randint(0, 256, (L, B, Hkv, T + 4, kd), dtype=torch.uint8, device="cuda", generator=g).view(K.QDTYPE[kinds[0]])
        vs_ = torch.randint(0, 256, (L, B, Hkv, T + 4, vd), dtype=torch.uint8, device="cuda", generator=g).view(K.QDTYPE[kinds[1]])
        if kinds[0] == "int8":
            ks_.clamp_(min=-127)
        if kinds[1] == "int8":
            vs_.clamp_(min=-127)
        ksc = torch.rand(L, T + 4, device="cuda", generator=g) * 0.02 + 1e-3
        vsc = torch.rand(L, T + 4, device="cuda", generator=g) * 0.2 + 1e-3
        q = torch.randn(L, B, Hq, D, device="cuda", generator=g).to(dtype)
        kn = torch.randn(L, B, Hkv, D, device="cuda", generator=g).to(dtype)
        vn = torch.randn(L, B, Hkv, D, device="cuda", generator=g).to(dtype)
        ws = torch.empty(K.decode_attn_workspace_cap(B, Hq, Hkv, T + 4, D), dtype=torch.float32, device="cuda")
        res = {}
        for fold in (0, 1, 2):
            tunable("attn_fold", fold)
            ws.fill_(float("nan"))  # all-ones exponent bit patterns in the arrival words too
            out = torch.full((L, B, Hq, D), float("nan"), dtype=dtype, device="cuda")
            plan = K.DecodeLayersPlan(q, kn, vn, out, ks_, ksc, kinds[0], vs_, vsc, kinds[1])
            _lib.kernel_log_clear()
            for _ in range(3):  # the words end every launch at zero: the next layer and the next call find them so
                K.decode_step_layers(plan, T, ws, D ** -0.5)
            torch.cuda.synchronize()
            log = _lib.kernel_log()
            assert log[0].startswith("decode_attn_lds_mfma_k<"), log
            assert len(log) == (2 if fold == 0 or Hq // Hkv > 4 else 1), (fold, log)
            assert torch.isfinite(out.float()).all()
            # single-layer entry point, without a new token: folded only with attn_fold = 2
            out1 = torch.full((B, Hq, D), float("nan"), dtype=dtype, device="cuda")
            _lib.kernel_log_clear()
            K.decode_attn(q[0], ks_[0], ksc[0], kinds[0], vs_[0], vsc[0], kinds[1], T, out1, ws, D ** -0.5)
            torch.cuda.synchronize()
            assert len(_lib.kernel_log()) == (1 if fold == 2 and Hq // Hkv <= 4 else 2), (fold, _lib.kernel_log())
            res[fold] = (out, out1)
        for fold in (1, 2):
            assert torch.equal(res[fold][0].view(torch.int16), res[0][0].view(torch.int16)), (dtype, kinds, fold)
            assert torch.equal(res[fold][1].view(torch.int16), res[0][1].view(torch.int16)), (dtype, kinds, fold, "no new token")


@pytest.mark.ab
def test_merge_inside_the_partial_launch_with_eight_query_heads_keeps_two_launches(K, tunable):
    from efficient_llm_inference_amd import _lib
    tunable("attn_stream_tpw", 2)
    tunable("attn_fold", 2)
    _lib.kernel_log_clear()
    _run_case(K, 1, 16, 2, 700, 128, "int8", "int4", "f16", True)
    assert len(_lib.kernel_log()) == 2


@pytest.mark.parametrize("tpw", [0, 1, 3, 9])
def test_decode_attn_lds_staged_kernel_head_dim_64(K, tunable, tpw):
    """decode_attn_lds_mfma_k<..., HD = 64> (round 4): the ring kernel at head_dim 64 with INT8 keys (64-byte key rows in the
    4-chunk swizzled image, 32- / 64-byte value rows, 2-byte V fragments for INT4 values) — Llama-3.2-1B / gpt2-family head
    shapes. Against the oracle: 4 query heads per kv head (the one-score-output instantiation), 8 and 3 per kv head (one per
    16-token group, padded heads), INT4 and INT8 values, bf16, with and without the new token, ragged tiles, a batch-8 context.
    INT4 KEYS at head_dim 64 have 32-byte rows (8-byte fragments): they keep the one-tile kernel."""
    from efficient_llm_inference_amd import _lib
    tunable("attn_stream_tpw", tpw)
    cases = [(1, 32, 8, 700, 64), (2, 16, 2, 131, 64), (1, 12, 4, 1000, 64), (3, 8, 2, 1, 64), (1, 8, 2, 64, 64), (1, 8, 2, 129, 64), (8, 32, 8, 4100, 64)]
    for case in cases:
        big = case[0] * case[3] > 20000
        if big and tpw not in (0, 3):
            continue
        _run_case(K, *case, "int8", "int4", "f16", True)
        if not big:
            _run_case(K, *case, "int8", "int8", "f16", True)
            _run_case(K, *case, "int8", "int4", "bf16", True)
            _run_case(K, *case, "int8", "int4", "f16", False)
    if tpw:
        _lib.kernel_log_clear()
        _run_case(K, 2, 32, 8, 1500, 64, "int8", "int4", "f16", True)
        assert _lib.kernel_log()[0].startswith("decode_attn_lds_mfma_k<8, 4, 64, true, 2, 4, 64, false>"), _lib.kernel_log()
        _lib.kernel_log_clear()
        _run_case(K, 1, 16, 2, 700, 64, "int8", "int8", "f16", True)  # 8 query heads per kv head
        assert _lib.kernel_log()[0].startswith("decode_attn_lds_mfma_k<8, 8, 64, true, 2, 1, 64, false>"), _lib.kernel_log()
        _lib.kernel_log_clear()
        _run_case(K, 2, 32, 8, 1500, 64, "int4", "int8", "f16", True)  # INT4 keys: not the ring
        assert _lib.kernel_log()[0].startswith("decode_attn_partial_mfma_k<4, 8, 128, 64"), _lib.kernel_log()


# ---------------------------------------------------------------------------- one pass (A-B key attn_onepass)

ONEPASS_CASES = [c for c in CASES if c[3] <= 2048 and c[4] in (64, 128) and c[1] // c[2] * c[4] <= 512] + [
    (1, 12, 12, 1024, 64), (1, 12, 12, 1025, 64), (2, 32, 8, 2048, 128), (1, 8, 8, 1500, 128), (8, 32, 8, 129, 128)]


@pytest.mark.ab
@pytest.mark.parametrize("case", ONEPASS_CASES)
@pytest.mark.parametrize("kinds", [("int8", "int8"), ("int8", "int4"), ("int4", "int4"), ("int4", "int8")])
def test_decode_attn_one_pass_matches_oracle(K, tunable, case, kinds):
    """A-B key attn_onepass: up to 2,048 stored tokens in ONE launch — one 8-wave workgroup per (batch row, kv head), one or
    two 128-token tiles per wave, merge of the <= 16 slots and of the exact new token from LDS. Same oracle, same tolerance as
    the two-launch path; the kernel log shows the single kernel."""
    from efficient_llm_inference_amd import _lib
    tunable("attn_onepass", 1)
    for dtype, with_new in (("f16", True), ("f16", False), ("bf16", True)):
        _lib.kernel_log_clear()
        _run_case(K, *case, kinds[0], kinds[1], dtype, with_new)
        log = _lib.kernel_log()
        assert len(log) == 1 and log[0].startswith("decode_attn_onepass_k<"), log


@pytest.mark.ab
@pytest.mark.parametrize("shape", [(1, 12, 12, 64), (2, 32, 8, 128)])
@pytest.mark.parametrize("kinds", [("int8", "int8"), ("int8", "int4")])
def test_one_pass_decode_step_with_a_device_side_token_count(K, tunable, shape, kinds):
    """The one-pass kernel behind kvq_decode_step_dev (what a captured decode graph replays): one set of launch arguments
    for every context length up to the bound — here a bound past 1,024 tokens, so two tiles per wave — attention within
    tolerance, slot T quantised bit-exactly by the launch's two extra workgroups, nothing else touched."""
    from efficient_llm_inference_amd import _lib
    tunable("attn_onepass", 1)
    B, Hq, Hkv, D = shape
    cap, bound = 1310, 1300
    rng = np.random.default_rng(Hq * 11 + D)
    k = rng.standard_normal((1, B, Hkv, cap, D)).astype(np.float16)
    v = rng.standard_normal((1, B, Hkv, cap, D)).astype(np.float16)
    kq, _, ks = O.quantize_tokens(k, kinds[0])
    vq, _, vs = O.quantize_tokens(v, kinds[1])
    q = rng.standard_normal((B, Hq, D)).astype(np.float16)
    sm = D ** -0.5
    qt = to_torch(q)
    ws = torch.empty(K.decode_attn_workspace_cap(B, Hq, Hkv, cap, D), dtype=torch.float32, device="cuda")
    t_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
    for T in (1, 127, 128, 129, 1023, 1024, 1025, 1299):
        k_store = torch.full((B, Hkv, cap, kq.shape[-1]), 9, dtype=K.QDTYPE[kinds[0]], device="cuda")
        v_store = torch.full((B, Hkv, cap, vq.shape[-1]), 9, dtype=K.QDTYPE[kinds[1]], device="cuda")
        k_sc = torch.full((cap,), -1.0, device="cuda")
        v_sc = torch.full((cap,), -1.0, device="cuda")
        k_store[:, :, :T] = to_torch(kq[0][:, :, :T])
        v_store[:, :, :T] = to_torch(vq[0][:, :, :T])
        k_sc[:T] = to_torch(ks[0][:T])
        v_sc[:T] = to_torch(vs[0][:T])
        kn, vn = to_torch(k[0][:, :, T].copy()), to_torch(v[0][:, :, T].copy())
        out = torch.full_like(qt, float("nan"))
        plan = K.DecodeStepPlan(qt, k_store, k_sc, kinds[0], v_store, v_sc, kinds[1], 1e-8)
        t_dev.fill_(T)
        _lib.kernel_log_clear()
        K.decode_step_dev(plan, qt, kn, vn, t_dev, bound, out, ws, sm)
        torch.cuda.synchronize()
        assert [n.split("<")[0] for n in _lib.kernel_log()] == ["decode_attn_onepass_k"], _lib.kernel_log()
        ref = O.decode_attention(q, kq[0][:, :, :T], ks[0][:T], kinds[0], vq[0][:, :, :T], vs[0][:T], kinds[1], D, sm,
                                 k[0][:, :, T].astype(np.float32), v[0][:, :, T].astype(np.float32))
        got = to_numpy(out).astype(np.float64)
        assert np.isfinite(got).all() and (np.abs(got - ref) <= TOL["f16"] * (np.abs(ref) + np.abs(ref).max())).all(), T
        assert np.array_equal(to_numpy(k_store[:, :, :T + 1]), kq[0][:, :, :T + 1]) and np.array_equal(to_numpy(v_store[:, :, :T + 1]), vq[0][:, :, :T + 1]), T
        assert np.array_equal(to_numpy(k_sc[:T + 1]), ks[0][:T + 1]) and np.array_equal(to_numpy(v_sc[:T + 1]), vs[0][:T + 1])
        assert int((k_store[:, :, T + 1:] != 9).sum()) == 0 and float((k_sc[T + 1:] + 1.0).abs().sum()) == 0.0


# ---------------------------------------------------------------------------- new-token slices past the register path

@pytest.mark.parametrize("shape,kinds,dtype", [
    ((16, 32, 8, 300, 128), ("int8", "int4"), "f16"),   # 2 pieces per slice
    ((64, 8, 8, 130, 128), ("int8", "int4"), "f16"),    # 8 pieces: the 65,536-element bound of the fused append
    ((64, 8, 8, 130, 128), ("int4", "int8"), "bf16"),
    ((33, 12, 12, 77, 64), ("int4", "int8"), "f16"),    # head_dim 64, 12 heads: pieces that end inside a row walk step
    ((9, 16, 8, 200, 128), ("int8", "int8"), "bf16"),   # 9,216 elements: just past the register path
])
def test_decode_step_appends_a_large_batch_with_several_workgroups(K, tunable, shape, kinds, dtype):
    """kvq_decode_step / kvq_decode_step_dev with a new-token slice of more than 8,192 elements (batch > 8 at 8 kv heads x 128):
    `quant_new_token_parts` — one workgroup per 8,192 elements, each taking the abs-max of the WHOLE slice and quantising its own
    piece — instead of ONE workgroup walking the slice element by element (158 us per layer at batch 64). Slot T and its scale
    bit-exact with the oracle's quantiser and with the one-workgroup routine (knob attn_new_token_parts = 0), nothing else
    touched, attention within tolerance; host-side and device-side token count."""
    B, Hq, Hkv, T, D = shape
    rng = np.random.default_rng(B * 131 + T)
    mk = (lambda s: O.f32_to_bf16_bits(rng.standard_normal(s).astype(np.float32))) if dtype == "bf16" else (lambda s: rng.standard_normal(s).astype(np.float16))
    k, v = mk((1, B, Hkv, T + 1, D)), mk((1, B, Hkv, T + 1, D))
    odt_ = "bf16" if dtype == "bf16" else None
    kq, _, ks = O.quantize_tokens(k, kinds[0], dtype=odt_)
    vq, _, vs = O.quantize_tokens(v, kinds[1], dtype=odt_)
    q = mk((B, Hq, D))
    sm = D ** -0.5
    ref = O.decode_attention(_as_f32(q, dtype), kq[0][:, :, :T], ks[0][:T], kinds[0], vq[0][:, :, :T], vs[0][:T], kinds[1], D, sm,
                             _as_f32(k[0][:, :, T], dtype), _as_f32(v[0][:, :, T], dtype), kv_dtype=dtype)
    cap = T + 5
    qt = to_torch(q, dtype)
    kn, vn = to_torch(np.ascontiguousarray(k[0][:, :, T]), dtype), to_torch(np.ascontiguousarray(v[0][:, :, T]), dtype)
    ws = torch.empty(K.decode_attn_workspace_cap(B, Hq, Hkv, cap, D), dtype=torch.float32, device="cuda")
    t_dev = torch.full((1,), T, dtype=torch.int32, device="cuda")
    stores = {}
    for knob, dev_side in ((1, False), (1, True), (0, False)):
        tunable("attn_new_token_parts", knob)
        k_store = torch.full((B, Hkv, cap, kq.shape[-1]), 9, dtype=K.QDTYPE[kinds[0]], device="cuda")
        v_store = torch.full((B, Hkv, cap, vq.shape[-1]), 9, dtype=K.QDTYPE[kinds[1]], device="cuda")
        k_sc = torch.full((cap,), -1.0, device="cuda")
        v_sc = torch.full((cap,), -1.0, device="cuda")
        k_store[:, :, :T] = to_torch(kq[0][:, :, :T])
        v_store[:, :, :T] = to_torch(vq[0][:, :, :T])
        k_sc[:T] = to_torch(ks[0][:T])
        v_sc[:T] = to_torch(vs[0][:T])
        out = torch.full_like(qt, float("nan"))
        plan = K.DecodeStepPlan(qt, k_store, k_sc, kinds[0], v_store, v_sc, kinds[1], 1e-8)
        if dev_side:
            K.decode_step_dev(plan, qt, kn, vn, t_dev, T + 2, out, ws, sm)
        else:
            K.decode_step(plan, qt, kn, vn, T, out, ws, sm)
        torch.cuda.synchronize()
        got = _as_f32(to_numpy(out), dtype).astype(np.float64)
        assert np.isfinite(got).all() and (np.abs(got - ref) <= TOL[dtype] * (np.abs(ref) + np.abs(ref).max())).all(), (knob, dev_side)
        assert np.array_equal(to_numpy(k_store[:, :, :T + 1]), kq[0]) and np.array_equal(to_numpy(v_store[:, :, :T + 1]), vq[0]), (knob, dev_side)
        assert np.array_equal(to_numpy(k_sc[:T + 1]), ks[0]) and np.array_equal(to_numpy(v_sc[:T + 1]), vs[0]), (knob, dev_side)
        assert int((k_store[:, :, T + 1:] != 9).sum()) == 0 and int((v_store[:, :, T + 1:] != 9).sum()) == 0
        assert float((k_sc[T + 1:] + 1.0).abs().sum()) == 0.0 and float((v_sc[T + 1:] + 1.0).abs().sum()) == 0.0
        stores[(knob, dev_side)] = (k_store, v_store)
    assert torch.equal(stores[(1, False)][0], stores[(0, False)][0]) and torch.equal(stores[(1, False)][1], stores[(0, False)][1])


# ---------------------------------------------------------------------------- device-side token count on the ring kernel

@pytest.mark.parametrize("shape,kinds", [((8, 32, 8, 128), ("int8", "int4")), ((8, 32, 8, 64), ("int8", "int8"))])
def test_decode_step_dev_on_the_ring_kernel(K, tunable, shape, kinds):
    """kvq_decode_step_dev at a batch and bound where the host-side call takes the LDS-staged ring kernel: since round 4 the
    device-side-count call takes it too — every wave derives its tile range from the count (ceil(tiles / nsplit) tiles per wave,
    waves past the count exit), the merge derives the live splits the same way. For a sweep of counts under ONE bound: the ring
    kernel is what runs, attention within tolerance of the oracle, slot T quantised bit-exactly, nothing else touched; the
    knob attn_ring_dev = 0 gives the one-tile splits of before, same tolerance."""
    from efficient_llm_inference_amd import _lib
    B, Hq, Hkv, D = shape
    cap, bound = 3110, 3100
    rng = np.random.default_rng(Hq * 13 + D)
    k = rng.standard_normal((1, B, Hkv, cap, D)).astype(np.float16)
    v = rng.standard_normal((1, B, Hkv, cap, D)).astype(np.float16)
    kq, _, ks = O.quantize_tokens(k, kinds[0])
    vq, _, vs = O.quantize_tokens(v, kinds[1])
    q = rng.standard_normal((B, Hq, D)).astype(np.float16)
    sm = D ** -0.5
    qt = to_torch(q)
    ws = torch.empty(K.decode_attn_workspace_cap(B, Hq, Hkv, cap, D), dtype=torch.float32, device="cuda")
    ws.fill_(float("nan"))  # stale partials of dead splits must never reach the output
    t_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
    kq_t, vq_t, ks_t, vs_t = to_torch(kq[0]), to_torch(vq[0]), to_torch(ks[0]), to_torch(vs[0])
    for knob, counts in ((1, (1, 64, 65, 2049, 3099)), (0, (3099,))):
        tunable("attn_ring_dev", knob)
        for T in counts:
            k_store = torch.full((B, Hkv, cap, kq.shape[-1]), 9, dtype=K.QDTYPE[kinds[0]], device="cuda")
            v_store = torch.full((B, Hkv, cap, vq.shape[-1]), 9, dtype=K.QDTYPE[kinds[1]], device="cuda")
            k_sc = torch.full((cap,), -1.0, device="cuda")
            v_sc = torch.full((cap,), -1.0, device="cuda")
            k_store[:, :, :T] = kq_t[:, :, :T]
            v_store[:, :, :T] = vq_t[:, :, :T]
            k_sc[:T] = ks_t[:T]
            v_sc[:T] = vs_t[:T]
            kn, vn = to_torch(k[0][:, :, T].copy()), to_torch(v[0][:, :, T].copy())
            out = torch.full_like(qt, float("nan"))
            plan = K.DecodeStepPlan(qt, k_store, k_sc, kinds[0], v_store, v_sc, kinds[1], 1e-8)
            t_dev.fill_(T)
            _lib.kernel_log_clear()
            K.decode_step_dev(plan, qt, kn, vn, t_dev, bound, out, ws, sm)
            torch.cuda.synchronize()
            log = _lib.kernel_log()
            assert (log[0].startswith("decode_attn_lds_mfma_k<") and log[0].endswith(", true>") and log[1].endswith(", true>")) if knob else log[0].startswith("decode_attn_partial_mfma_k<"), (knob, T, log)  # (the ring's and the merge's device-count instantiations)
            ref = O.decode_attention(q, kq[0][:, :, :T], ks[0][:T], kinds[0], vq[0][:, :, :T], vs[0][:T], kinds[1], D, sm,
                                     k[0][:, :, T].astype(np.float32), v[0][:, :, T].astype(np.float32))
            got = to_numpy(out).astype(np.float64)
            assert np.isfinite(got).all() and (np.abs(got - ref) <= TOL["f16"] * (np.abs(ref) + np.abs(ref).max())).all(), (knob, T)
            assert torch.equal(k_store[:, :, T], kq_t[:, :, T]) and torch.equal(v_store[:, :, T], vq_t[:, :, T]), (knob, T)
            assert float(k_sc[T]) == float(ks_t[T]) and float(v_sc[T]) == float(vs_t[T])
            assert int((k_store[:, :, T + 1:] != 9).sum()) == 0 and float((k_sc[T + 1:] + 1.0).abs().sum()) == 0.0
