"""Property-based checks (hypothesis) of the CPU oracles over random shapes, incl. odd D, zeros,
single elements, T <= window and ragged chunks. CPU only; sized to run in seconds."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import c_oracle as C
from oracle import kvq_oracle as O

shapes = st.tuples(st.integers(1, 3), st.integers(1, 3), st.integers(1, 4), st.integers(1, 9), st.integers(1, 17))


def _kv(shape, seed, scale, dtype):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(shape) * scale).astype(np.float32)
    x[rng.random(shape) < 0.1] = 0.0
    return x.astype(dtype)


@settings(max_examples=60, deadline=None)
@given(shapes, st.integers(0, 2**31 - 1), st.sampled_from([1e-6, 1.0, 300.0]), st.sampled_from([np.float16, np.float32]))
def test_pack_unpack_and_error_bound(shape, seed, scale, dtype):
    x = _kv(shape, seed, scale, dtype)
    D = shape[-1]
    for kind, qmax in (("int8", 127.0), ("int4", 7.0)):
        q, stored, s32 = O.quantize_tokens(x, kind)
        qi = O.unpack_int4(q, D) if kind == "int4" else q
        assert qi.min() >= -qmax and qi.max() <= qmax  # -8 is never produced
        if kind == "int4":
            assert np.array_equal(O.pack_int4(qi), q)  # pack . unpack = identity (pad nibble = 8)
            if D % 2:
                assert np.all((q[..., -1] & 0x0F) == 8)
        # reconstruction: |x - q*s| <= s/2 per token (s = the fp32 scale used to quantise)
        s_used = np.maximum(np.abs(x.astype(np.float32)).max(axis=(1, 2, 4)) / np.float32(qmax), np.float32(1e-8))
        x64, s64 = x.astype(np.float64), s_used.astype(np.float64)[:, None, None, :, None]
        err = np.abs(x64 - qi.astype(np.float64) * s64)
        # half a step, plus the fp32 rounding of the quotient x/s before rint (<= 2^-24 |x/s| * s)
        assert np.all(err <= 0.5 * s64 + np.abs(x64) * 2.0**-23 + 1e-30)
        # the token holding max|x| quantises to +-qmax whenever the scale is not the eps floor
        big = s_used > 1e-8
        assert np.all(np.abs(qi).max(axis=(1, 2, 4))[big] == qmax)
        # C oracle agrees bit for bit
        qc, sc = C.quantize_tokens(x, kind)
        assert np.array_equal(qc, q) and np.array_equal(sc.view(np.uint32), s32.view(np.uint32))
        for od in ("f16", "f32"):
            assert np.array_equal(C.dequantize_tokens(qc, sc, kind, D, od).view(np.uint8),
                                  O.dequantize_tokens(q, s32, kind, D, od).view(np.uint8))


@settings(max_examples=80, deadline=None)
@given(st.integers(1, 200), st.integers(1, 70), st.integers(0, 80), st.integers(0, 2**31 - 1))
def test_chunk_summary_properties(T, chunk, keep, seed):
    x = _kv((2, T, 4), seed, 1.0, np.float32)
    out = O.chunk_summarize_kv(x, chunk, keep)
    assert out.shape[-2] == O.chunk_summary_len(T, chunk, keep)
    k = min(keep, T)
    old = T - k
    if old <= 0:
        assert out is x
        return
    assert np.array_equal(out[..., out.shape[-2] - k:, :], x[..., old:, :])  # kept tail: exact copy
    n = (old + chunk - 1) // chunk
    # mass conservation: chunk_size * sum(summaries) = sum(old tokens) (divisor is chunk_size even
    # for the ragged last chunk, implementations.py:326-339)
    assert np.allclose(out[..., :n, :].sum(axis=-2, dtype=np.float64) * chunk, x[..., :old, :].sum(axis=-2, dtype=np.float64),
                       rtol=1e-4, atol=1e-4)
    assert np.array_equal(C.chunk_summarize(x, chunk, keep).view(np.uint8), out.view(np.uint8))


@settings(max_examples=150, deadline=None)
@given(st.integers(1, 400), st.integers(1, 64), st.integers(0, 40), st.integers(1, 8), st.integers(1, 32),
       st.integers(1, 16), st.integers(0, 70))
def test_sparse_policies_structure(T, W, P, stride, bs, kpb, budget):
    kpb = min(kpb, bs)
    lists = [O.keep_indices_prefix_window(T, P, W), O.keep_indices_strided(T, W, stride, P),
             O.keep_indices_block_old(T, W, bs, kpb, P), O.keep_indices_budget_old(T, W, budget, P)]
    for idx in lists:
        if T <= P + W:
            assert idx is None
            continue
        ts = max(P, T - W)
        assert idx[:P] == list(range(P)) and idx[len(idx) - (T - ts):] == list(range(ts, T))  # prefix + dense tail
        assert all(0 <= i < T for i in idx) and idx == sorted(idx) and len(set(idx)) == len(idx)
    if T > P + W:
        ts = max(P, T - W)
        assert len(lists[3]) <= P + min(budget, ts - P) + (T - ts)
        assert len(lists[1]) == P + len(range(P, ts, stride)) + (T - ts)


@settings(max_examples=40, deadline=None)
@given(st.tuples(st.integers(1, 3), st.integers(2, 6), st.integers(1, 4), st.integers(1, 9), st.integers(1, 17)),
       st.integers(0, 2**31 - 1), st.sampled_from([1e-6, 1.0, 300.0]), st.sampled_from([np.float16, np.float32]),
       st.integers(1, 5))
def test_split_phase_quantise_is_shard_invariant(shape, seed, scale, dtype, cut):
    """SURVEY §8e's exchange step on the oracle: for ANY split of the batch rows, the element-wise MAX of the
    shards' abs-max tables is the whole batch's table, and quantising every shard with it reproduces the
    un-sharded quantise (q and scales bit for bit) — what the sharded HIP path relies on."""
    x = _kv(shape, seed, scale, dtype)
    B = shape[1]
    cut = min(cut, B - 1)
    shards = [x[:, :cut], x[:, cut:]]
    amax = np.maximum(O.absmax_tokens(shards[0]), O.absmax_tokens(shards[1]))
    assert np.array_equal(amax.view(np.uint32), O.absmax_tokens(x).view(np.uint32))
    for kind in ("int8", "int4"):
        q_ref, st_ref, s32_ref = O.quantize_tokens(x, kind)
        parts = [O.quantize_tokens_with_absmax(sh, amax, kind) for sh in shards]
        assert np.array_equal(np.concatenate([p[0] for p in parts], axis=1), q_ref)
        for p in parts:
            assert np.array_equal(p[2].view(np.uint32), s32_ref.view(np.uint32))
            assert np.array_equal(np.asarray(p[1]).view(np.uint8), np.asarray(st_ref).view(np.uint8))
