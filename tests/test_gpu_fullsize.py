"""Full-size checks at BASELINE.json's shapes (too large for the CPU oracle): size-independent
properties + an independent re-computation of the reference's formulas with plain torch ops on
the GPU (the reference's own fallback expressions, ops.py:26-30, 47-63, 90, 122-133), layer by
layer. Run on a real MI355X: ``pytest -m gpu``.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

LLAMA = (32, 1, 8, 16384, 128)  # config 4: Llama-3-8B, seq 16K   [L,B,H,T,D]
GPT2M = (24, 1, 16, 4096, 64)   # config 3: gpt2-medium, seq 4K


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available()
    from efficient_llm_inference_amd import _lib, kernels
    _lib.load()
    return kernels


def _torch_unpack(p):  # ops.py:122-131
    hi = (p >> 4) & 0x0F
    lo = p & 0x0F
    return torch.stack([hi, lo], dim=-1).flatten(-2).to(torch.int16) - 8


def _torch_quant(x, qmax, qmin):  # ops.py:26-29 / 47-50 per token slice, vectorised over t
    x32 = x.float()
    amax = x32.abs().amax(dim=(0, 1, 3))  # [T] over B,H,D
    # The scale is computed on the CPU: torch's GPU kernel for tensor / python-scalar multiplies by
    # fl(1/qmax) instead of dividing (<= 1 ulp off the true quotient). The oracle — the
    # reference as it runs in the build container, on CPU — divides; so does the HIP kernel.
    s32 = (amax.cpu() / qmax).clamp(min=1e-8).to(x.device)
    q = torch.clamp((x32 / s32[None, None, :, None]).round(), qmin, qmax).to(torch.int8)
    return q, s32


@pytest.mark.parametrize("shape", [LLAMA, GPT2M])
@pytest.mark.parametrize("kind", ["int4", "int8"])
def test_fullsize_dequant_matches_reference_formula(K, shape, kind):
    L, B, H, T, D = shape
    g = torch.Generator(device="cuda").manual_seed(1)
    Dq = K.packed_dim(kind, D)
    if kind == "int4":
        q = torch.randint(0, 256, (L, B, H, T, Dq), dtype=torch.uint8, device="cuda", generator=g)
    else:
        q = torch.randint(-127, 128, (L, B, H, T, Dq), dtype=torch.int8, device="cuda", generator=g)
    scales = (torch.rand(L, T, device="cuda", generator=g) * 0.05).half().float()  # stored fp16 scales, widened
    scales[:, ::97] = 0.0  # underflowed scales: exercises the -0.0 results
    out = torch.empty(L, B, H, T, D, dtype=torch.float16, device="cuda")
    K.dequant_tokens(q, scales, out, kind)
    for l in range(L):
        qi = _torch_unpack(q[l]) if kind == "int4" else q[l]
        ref = (qi.float() * scales[l][None, None, :, None]).half()
        assert torch.equal(out[l].view(torch.int16), ref.view(torch.int16)), f"layer {l}"


@pytest.mark.parametrize("shape", [LLAMA, GPT2M])
@pytest.mark.parametrize("kind", ["int4", "int8"])
def test_fullsize_quant_roundtrip(K, shape, kind):
    """quantise -> dequantise at full size: packed bytes / int8 and stored scales equal the torch
    re-computation; reconstruction error within half a quantisation step (+ fp16 rounding)."""
    L, B, H, T, D = shape
    qmax, qmin = (7.0, -8.0) if kind == "int4" else (127.0, -127.0)
    g = torch.Generator(device="cuda").manual_seed(2)
    Dq = K.packed_dim(kind, D)
    store = torch.empty(L, B, H, T, Dq, dtype=K.QDTYPE[kind], device="cuda")
    scales = torch.empty(L, T, dtype=torch.float32, device="cuda")
    ws = torch.empty(L * T, dtype=torch.float32, device="cuda")
    xs = []
    for l in range(L):  # separately allocated tensors: the pointer-table launch
        x = torch.randn(B, H, T, D, device="cuda", generator=g)
        x = torch.where(torch.rand(B, H, T, D, device="cuda", generator=g) < 0.01, x * 8, x).half()
        xs.append(x)
    K.quant_tokens(xs, store, scales, ws, kind)
    out = torch.empty(L, B, H, T, D, dtype=torch.float16, device="cuda")
    K.dequant_tokens(store, scales, out, kind)
    for l in range(L):
        q_ref, s32 = _torch_quant(xs[l], qmax, qmin)
        s_stored = s32.half().float()
        assert torch.equal(scales[l], s_stored), f"scales layer {l}"
        got = _torch_unpack(store[l]).to(torch.int8) if kind == "int4" else store[l]
        assert torch.equal(got, q_ref), f"q layer {l}"
        # |x - deq| <= s/2 (rounding) + qmax*|s - s_stored| (stored-scale rounding) + fp16 ulp of the result
        err = (xs[l].float() - out[l].float()).abs()
        bound = (0.5 * s32 + qmax * (s32 - s_stored).abs())[None, None, :, None] + out[l].float().abs() * 2.0**-10 + 1e-7
        assert bool((err <= bound).all()), f"round trip layer {l}"
    # idempotence of the representation: dequantising twice gives identical bytes
    out2 = torch.empty_like(out)
    K.dequant_tokens(store, scales, out2, kind)
    assert torch.equal(out.view(torch.int16), out2.view(torch.int16))


def test_fullsize_eviction_properties(K):
    """Config-5-shaped (per-GPU slice, fewer layers): window compaction is an exact gather;
    chunk mean-pool is linear: pool(a) + pool(b) ~= pool(a + b) and equals torch's mean within
    1 fp16 ulp; the kept tail is an exact copy."""
    G, B, H, T, D = 4, 8, 8, 32768, 128
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(G, B, H, T, D, device="cuda", generator=g).half()
    W = 256
    out = torch.empty(G, B, H, W, D, dtype=torch.float16, device="cuda")
    K.window_compact(x, out, W)
    assert torch.equal(out, x[:, :, :, T - W:])
    chunk, keep = 64, 256
    Tout = K.chunk_summary_len(T, chunk, keep)
    assert Tout == 764
    pooled = torch.empty(G, B, H, Tout, D, dtype=torch.float16, device="cuda")
    K.chunk_meanpool(x, pooled, chunk, keep)
    assert torch.equal(pooled[:, :, :, Tout - keep:], x[:, :, :, T - keep:])
    ref = x[:, :, :, : T - keep].float().view(G, B, H, -1, chunk, D).mean(dim=4)
    got = pooled[:, :, :, : Tout - keep].float()
    assert bool(((got - ref).abs() <= ref.abs() * 2.0**-10 + 1e-5).all())


def test_config5_per_gpu_share_all_64_tensors():
    """BASELINE configs[4] at its stated per-GPU size: Llama-3-8B sliding_window + chunk_summary, seq 32K,
    batch 64 over 8 GPUs = 8 batch rows per GPU: the legacy tuple of ALL 32 layers x (K, V) = 64 separately
    allocated [8, 8, 32768, 128] fp16 tensors (32 GiB) through the public trim_kv_sliding_window /
    chunk_summarize_kv (two launches each: 64 base pointers per launch). Window = exact gather of the last
    256 tokens; kept tail = exact copy; every pooled row within 1 fp16 ulp of torch's per-tensor mean."""
    import efficient_llm_inference_amd as E
    L, B, H, T, D = 32, 8, 8, 32768, 128
    W, chunk, keep = 256, 64, 256  # CacheConfig defaults (reference src/core/config.py:64-67)
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * 2**30:
        pytest.skip(f"needs 40 GiB of free HBM, {free / 2**30:.0f} GiB available")
    g = torch.Generator(device="cuda").manual_seed(5)
    past = []
    for _ in range(L):
        k = torch.empty(B, H, T, D, device="cuda", dtype=torch.float16).normal_(generator=g)
        v = torch.empty(B, H, T, D, device="cuda", dtype=torch.float16).normal_(generator=g)
        past.append((k, v))
    past = tuple(past)
    win = E.trim_kv_sliding_window(past, W)
    pooled = E.chunk_summarize_kv(past, chunk_size=chunk, keep_last=keep)
    torch.cuda.synchronize()
    Tout = 764  # 508 summaries + 256 recent
    assert len(win) == len(pooled) == L
    for l in range(L):
        for i in range(2):
            x, w, p = past[l][i], win[l][i], pooled[l][i]
            assert tuple(w.shape) == (B, H, W, D) and tuple(p.shape) == (B, H, Tout, D) and p.dtype == x.dtype
            assert torch.equal(w, x[:, :, T - W:]), f"window layer {l} tensor {i}"
            assert torch.equal(p[:, :, Tout - keep:], x[:, :, T - keep:]), f"tail layer {l} tensor {i}"
            ref = x[:, :, : T - keep].float().view(B, H, -1, chunk, D).mean(dim=3)
            got = p[:, :, : Tout - keep].float()
            assert bool(((got - ref).abs() <= ref.abs() * 2.0**-10 + 1e-5).all()), f"pool layer {l} tensor {i}"
            del ref, got


def test_timed_launch_binds_events_to_the_dispatch(K):
    """_lib.timed_launch (kvq_time_next_launch): the events carry the launch's own start / stop timestamps — positive,
    no longer than an event-record bracket around the same launch, one-shot (the following launch is a plain one),
    without effect on the result, disarmed when the block raises before the library is reached, and the same for any
    entry point (quantise, pool); kernel_log names the kernel that ran."""
    from efficient_llm_inference_amd import _lib
    L, B, H, T, D = LLAMA
    g = torch.Generator(device="cuda").manual_seed(7)
    q = torch.randint(0, 256, (L, B, H, T, D // 2), dtype=torch.uint8, device="cuda", generator=g)
    s = torch.rand(L, T, device="cuda", generator=g) * 0.01 + 1e-4
    ref = torch.empty(L, B, H, T, D, dtype=torch.float16, device="cuda")
    out = torch.empty_like(ref)
    _lib.kernel_log_clear()
    K.dequant_tokens(q, s, ref, "int4")
    assert _lib.kernel_log() == ["dequant_tokens_fast_k<0, 4, 8, 4, true, true, false, 64>"], _lib.kernel_log()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for e in ev:
        e.record()
    torch.cuda.synchronize()
    bound, bracket = [], []
    for _ in range(5):
        out.zero_()
        ev[2].record()
        with _lib.timed_launch(ev[0], ev[1]):
            K.dequant_tokens(q, s, out, "int4")
        ev[3].record()
        torch.cuda.synchronize()
        bound.append(ev[0].elapsed_time(ev[1]))
        bracket.append(ev[2].elapsed_time(ev[3]))
        assert torch.equal(out, ref)
    b, r = min(bound), min(bracket)
    assert 0.05 < b <= r * 1.02, (bound, bracket)  # 0.8 GiB moved: > 50 us on any HBM; never longer than the bracket
    # one-shot: a second launch does not touch the events
    before = ev[0].elapsed_time(ev[1])
    K.dequant_tokens(q, s, out, "int4")
    torch.cuda.synchronize()
    assert ev[0].elapsed_time(ev[1]) == before
    # the element-wise kernel (head_dim not a multiple of 8) is bound the same way
    q8 = torch.randint(-127, 128, (2, 1, 4, 4096, 100), dtype=torch.int8, device="cuda", generator=g)
    s8 = torch.rand(2, 4096, device="cuda", generator=g) * 0.01 + 1e-4
    o8 = torch.empty(2, 1, 4, 4096, 100, dtype=torch.float16, device="cuda")
    with _lib.timed_launch(ev[0], ev[1]):
        K.dequant_tokens(q8, s8, o8, "int8")
    torch.cuda.synchronize()
    small = ev[0].elapsed_time(ev[1])
    assert 0.0 < small < before
    # a call that raises before it reaches the library (shape validation) leaves nothing armed: the launch after the
    # block is a plain one (stale handles must never bind to a later, unrelated launch)
    with pytest.raises(RuntimeError):
        with _lib.timed_launch(ev[0], ev[1]):
            K.dequant_tokens(q8, s8, o8[:, :, :, :7], "int8")
    K.dequant_tokens(q, s, out, "int4")
    torch.cuda.synchronize()
    assert ev[0].elapsed_time(ev[1]) == small
    # an empty table launches nothing; the pair is disarmed by the block's exit all the same
    with _lib.timed_launch(ev[0], ev[1]):
        K.dequant_tokens(q8[:, :, :, :0], s8[:, :0], o8[:, :, :, :0], "int8")
    K.dequant_tokens(q, s, out, "int4")
    torch.cuda.synchronize()
    assert ev[0].elapsed_time(ev[1]) == small
    # any entry point: the quantise launch of the Llama shape is the compile-time tile kernel
    x = torch.randn(4, B, H, 4096, D, device="cuda", dtype=torch.float16, generator=g)
    qs = torch.empty(4, B, H, 4096, D // 2, dtype=torch.uint8, device="cuda")
    sc = torch.empty(4, 4096, dtype=torch.float32, device="cuda")
    _lib.kernel_log_clear()
    with _lib.timed_launch(ev[0], ev[1]):
        K.quant_tokens(x, qs, sc, torch.empty(4 * 4096, dtype=torch.float32, device="cuda"), "int4")
    torch.cuda.synchronize()
    assert 0.0 < ev[0].elapsed_time(ev[1]) < before
    assert _lib.kernel_log() == ["quant_tile_k<0, 4, 8, 16, 4, 0>"], _lib.kernel_log()


def test_context_of_128k_tokens_quantise_dequantise_and_attend(K):
    """Eight times the benchmark context (T = 131,072; Llama-3-8B row shape, 2 layers): the token-table kernels against
    the reference's formulas recomputed with torch ops layer by layer, and the decode attention (streaming kernel by
    size at batch 1, 2,048 tiles per kv head) against float64 attention over the dequantised store."""
    G, B, H, T, D = 2, 1, 8, 131072, 128
    g = torch.Generator(device="cuda").manual_seed(131072)
    x = torch.randn(G, B, H, T, D, device="cuda", generator=g).half()
    x[:, :, :, ::97, 5] *= 9.0
    for kind, qmax, qmin in (("int8", 127.0, -127.0), ("int4", 7.0, -8.0)):
        Dq = D if kind == "int8" else D // 2
        q = torch.empty(G, B, H, T, Dq, dtype=K.QDTYPE[kind], device="cuda")
        sc = torch.empty(G, T, dtype=torch.float32, device="cuda")
        ws = torch.empty(G * T, dtype=torch.float32, device="cuda")
        K.quant_tokens(x, q, sc, ws, kind)
        out = torch.empty(G, B, H, T, D, dtype=torch.float16, device="cuda")
        K.dequant_tokens(q, sc, out, kind)
        for l in range(G):
            qi, s32 = _torch_quant(x[l], qmax, qmin)
            stored = s32.half().float()
            assert torch.equal(sc[l], stored)
            ints = q[l].view(torch.int8).to(torch.int16) if kind == "int8" else _torch_unpack(q[l])
            assert torch.equal(ints, qi.to(torch.int16))
            ref = (qi.float() * stored[None, None, :, None]).half()
            assert torch.equal(out[l].view(torch.int16), ref.view(torch.int16))
    # attention over layer 0 of the INT4 store (values) and a fresh INT8 store (keys)
    kq = torch.empty(1, B, H, T, D, dtype=torch.int8, device="cuda")
    ks = torch.empty(1, T, dtype=torch.float32, device="cuda")
    K.quant_tokens(x[1:2], kq, ks, torch.empty(T, dtype=torch.float32, device="cuda"), "int8")
    Hq = 32
    qv = torch.randn(B, Hq, D, device="cuda", generator=g).half()
    o = torch.full((B, Hq, D), float("nan"), dtype=torch.float16, device="cuda")
    wsa = torch.empty(K.decode_attn_workspace(B, Hq, H, T, D), dtype=torch.float32, device="cuda")
    K.decode_attn(qv, kq[0], ks[0], "int8", q[0], sc[0], "int4", T, o, wsa, D ** -0.5, None, None)
    kd = torch.empty(1, B, H, T, D, dtype=torch.float16, device="cuda")
    K.dequant_tokens(kq, ks, kd, "int8")
    kf, vf = kd[0, 0].double(), out[0, 0].double()
    s = torch.einsum("hqd,htd->hqt", qv[0].double().view(H, Hq // H, D), kf) * D ** -0.5
    ref = torch.einsum("hqt,htd->hqd", torch.softmax(s, dim=-1), vf).reshape(Hq, D)
    got = o[0].double()
    assert torch.isfinite(got).all()
    assert bool(((got - ref).abs() <= 2e-3 * (ref.abs() + ref.abs().max())).all())
