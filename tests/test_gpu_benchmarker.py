"""End-to-end decode loops on the MI355X vs a reference-style emulation whose cache arithmetic is
the CPU oracle: because the HIP path is bit-exact, the model sees identical KV and must emit
identical tokens. BASELINE config 2 in miniature (random-init GPT-2 family, offline)."""
import numpy as np
import pytest
import torch

from oracle import kvq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rig():
    assert torch.cuda.is_available()
    from efficient_llm_inference_amd import KVCacheBenchmarker
    from efficient_llm_inference_amd.benchmarking import from_legacy_tuple, to_legacy_tuple
    from efficient_llm_inference_amd.benchmarking.offline import load_model
    model, tok = load_model("gpt2-tiny", "cuda", torch.float16)
    return KVCacheBenchmarker(model, tok, device="cuda"), model, tok, from_legacy_tuple, to_legacy_tuple


def _emulate(model, ids, n_new, policy, from_legacy, to_legacy):
    """The reference's loop shape (benchmarker.py:441-486) with `policy(kv_tuple) -> kv_tuple`
    evaluated by the oracle on the host."""
    out = model(input_ids=ids, use_cache=True)
    logits = out.logits[:, -1, :]
    past = policy(to_legacy(out.past_key_values), first=True)
    gen = ids.clone()
    for _ in range(n_new):
        nxt = torch.argmax(logits, dim=-1, keepdim=True)
        gen = torch.cat([gen, nxt], dim=-1)
        out = model(input_ids=nxt, use_cache=True, past_key_values=from_legacy(past))
        logits = out.logits[:, -1, :]
        past = policy(to_legacy(out.past_key_values), first=False)
    return gen


@pytest.mark.parametrize("mode", ["int8", "int4", "mixed"])
def test_quantized_decode_matches_oracle_emulation(rig, mode):
    bench, model, tok, from_legacy, to_legacy = rig
    kinds = {"int8": ("int8", "int8"), "int4": ("int4", "int4"), "mixed": ("int8", "int4")}[mode]
    store = {}

    def policy(kv, first):
        # quantise what is new (all prompt tokens first, then the last token), keep q + scales,
        # return the full dequantised cache: ops.py:323-355 semantics through the oracle
        outs = []
        for l, pair in enumerate(kv):
            both = []
            for i, kind in enumerate(kinds):
                x = pair[i].cpu().numpy()[None]  # [1,B,H,T,D]
                new = x if first else x[:, :, :, -1:]
                q, _, s32 = O.quantize_tokens(new, kind)
                if first:
                    store[(l, i)] = (q, s32)
                else:
                    q0, s0 = store[(l, i)]
                    store[(l, i)] = (np.concatenate([q0, q], axis=3), np.concatenate([s0, s32], axis=1))
                q, s32 = store[(l, i)]
                both.append(torch.from_numpy(O.dequantize_tokens(q, s32, kind, x.shape[-1], "f16")[0]).cuda())
            outs.append(tuple(both))
        return tuple(outs)

    with torch.no_grad():
        ids = tok("<37>", return_tensors="pt").input_ids.cuda()
        want = _emulate(model, ids, 12, policy, from_legacy, to_legacy)
        text, n_new, est_mb = bench.generate_with_quantized_kv("<37>", 12, mode=mode)
    assert n_new == 12 and text == tok.decode(want[0])
    cfg = model.config
    want_bytes = O.estimated_bytes(mode, cfg.n_layer, 1, cfg.n_head, 37 + 12, cfg.n_embd // cfg.n_head, 2)
    assert abs(est_mb - want_bytes / 2**20) < 1e-9


def test_sliding_window_and_chunked_decode_match_emulation(rig):
    bench, model, tok, from_legacy, to_legacy = rig
    with torch.no_grad():
        ids = tok("<60>", return_tensors="pt").input_ids.cuda()
        win = lambda kv, first: tuple((k[:, :, -16:, :].contiguous(), v[:, :, -16:, :].contiguous()) for k, v in kv)
        want = _emulate(model, ids, 10, win, from_legacy, to_legacy)
        text, n_new = bench.generate_with_sliding_window("<60>", 10, window_size=16)
        assert n_new == 10 and text == tok.decode(want[0])

        def pool(kv, first):
            return tuple(tuple(torch.from_numpy(O.chunk_summarize_kv(t.cpu().numpy(), 8, 12)).cuda() for t in pair)
                         for pair in kv)
        want = _emulate(model, ids, 10, pool, from_legacy, to_legacy)
        text, n_new, est_mb = bench.generate_with_chunked_cache("<60>", 10, chunk_size=8, keep_last=12)
        assert n_new == 10 and text == tok.decode(want[0]) and est_mb > 0


def test_benchmark_method_dict_on_gpu(rig):
    bench = rig[0]
    for method in ("full_cache", "quant_int8", "quant_int4", "quant_mixed", "sliding_window", "chunked_cache"):
        res = bench.benchmark_method(["<50>", "<70>"], method=method, max_new_tokens=6, window_size=32,
                                     chunk_size=8, keep_last=16)
        assert res["method"] == method and res["total_new_tokens"] == 12 and res["tokens_per_sec"] > 0
        assert res["gpu_peak_mb"] is not None
        if method.startswith("quant") or method == "chunked_cache":
            assert res["est_kv_cache_mb_avg"] > 0


@pytest.mark.parametrize("mode", ["int8", "int4", "mixed"])
def test_inplace_decode_equals_tuple_path(rig, mode):
    """The in-place staged HF cache (O(1) cache work per step) and the reference-shaped loop
    (dequantise everything, rebuild the cache, cat) generate the same tokens and report the same
    cache size."""
    bench = rig[0]
    try:
        bench.inplace_decode = True
        a = bench.generate_with_quantized_kv("<45>", 20, mode=mode)
        bench.inplace_decode = False
        b = bench.generate_with_quantized_kv("<45>", 20, mode=mode)
    finally:
        bench.inplace_decode = True
    assert a == b and a[1] == 20


def test_staged_cache_grows_without_reserve(rig):
    """Capacity not reserved: the staging buffers are reallocated mid-decode and the model's layers
    are re-bound; values stay those of a fresh full dequantise."""
    import efficient_llm_inference_amd as E
    from efficient_llm_inference_amd.benchmarking import to_legacy_tuple
    from efficient_llm_inference_amd.quantization import hf_cache
    bench, model, tok = rig[0], rig[1], rig[2]
    with torch.no_grad():
        ids = tok("<19>", return_tensors="pt").input_ids.cuda()
        out = model(input_ids=ids, use_cache=True)
        kv = to_legacy_tuple(out.past_key_values)
        qc = E.QuantizedKVCache(len(kv), "mixed")  # no reserve(): capacity 19 -> grows on the first append
        qc.init_from_prompt_past(kv)
        staged = hf_cache.StagedQuantizedCache(qc)
        logits = out.logits[:, -1, :]
        for _ in range(30):
            nxt = torch.argmax(logits, dim=-1, keepdim=True)
            out = model(input_ids=nxt, use_cache=True, past_key_values=staged.sync())
            logits = out.logits[:, -1, :]
            staged.commit()
        cache = staged.sync()
        fresh = E.QuantizedKVCache(len(kv), "mixed", incremental=False)
        fresh._k, fresh._v = qc._k, qc._v
        k_full = qc._k.dequant(torch.float16)
        assert len(qc.layers[0]) == 49 and cache.layers[0].get_seq_length() == 49
        for i, layer in enumerate(cache.layers):
            assert torch.equal(layer.keys, k_full[i]) and layer.keys.shape[-2] == 49


def test_bf16_model_decodes_with_bf16_compute():
    """Llama-family dtype: a bf16 model keeps bf16 as the cache's compute dtype (the reference's
    fp16-by-device-string choice would fail inside HF attention); both decode paths agree."""
    from efficient_llm_inference_amd import KVCacheBenchmarker
    from efficient_llm_inference_amd.benchmarking.offline import load_model
    model, tok = load_model("gpt2-tiny", "cuda", torch.bfloat16)
    b = KVCacheBenchmarker(model, tok, device="cuda")
    b.inplace_decode = False
    a1 = b.generate_with_quantized_kv("<33>", 8, "mixed")
    b.inplace_decode = True
    a2 = b.generate_with_quantized_kv("<33>", 8, "mixed")
    assert a1 == a2 and a1[1] == 8
    res = b.benchmark_method(["<40>"], "paged_attention", max_new_tokens=4, block_size=8)
    assert res["total_new_tokens"] == 4


@pytest.mark.parametrize("mode", ["int8", "mixed", "int4"])
def test_fused_attention_decode_tracks_tuple_path(mode):
    """Scope row N1, second form: the model attends straight over the INT8 / INT4 store
    (kvq_decode_attn as its attention function, no fp16 copy of the cache). Teacher-forced with the
    tuple path's tokens, every step's logits agree within fp16 tolerance; layer 0's store (a
    function of the embeddings only) is bit-identical; the reported cache size is the same."""
    import efficient_llm_inference_amd as E
    from efficient_llm_inference_amd import KVCacheBenchmarker
    from efficient_llm_inference_amd.benchmarking import from_legacy_tuple, to_legacy_tuple
    from efficient_llm_inference_amd.benchmarking.offline import load_model
    from efficient_llm_inference_amd.quantization import fused_attention as FA
    model, tok = load_model("gpt2-mini", "cuda", torch.float16)
    n_new = 24
    with torch.no_grad():
        ids = tok("<150>", return_tensors="pt").input_ids.cuda()
        out = model(input_ids=ids, use_cache=True)
        logits = out.logits[:, -1, :]
        kv = to_legacy_tuple(out.past_key_values)
        qc = E.QuantizedKVCache(len(kv), mode, incremental=False)
        qc.init_from_prompt_past(kv)
        first_logits = logits.float()
        ref_logits, toks = [], []
        for _ in range(n_new):
            nxt = torch.argmax(logits, dim=-1, keepdim=True)
            toks.append(nxt)
            out = model(input_ids=nxt, use_cache=True, past_key_values=from_legacy_tuple(qc.to_past_key_values()))
            logits = out.logits[:, -1, :]
            qc.append_from_past(to_legacy_tuple(out.past_key_values))
            ref_logits.append(logits.float())

        fc = FA.FusedQuantizedCache(len(kv), mode=mode, reserve=ids.shape[-1] + n_new)
        impl_before = model.config._attn_implementation
        with FA.fused_attention(model, fc) as cache:
            out = model(input_ids=ids, use_cache=True, past_key_values=cache)
            assert torch.allclose(out.logits[:, -1, :].float(), first_logits, atol=2e-2, rtol=0)  # exact prompt attention
            worst = 0.0
            for step, nxt in enumerate(toks):
                out = model(input_ids=nxt, use_cache=True, past_key_values=cache)
                got = out.logits[:, -1, :].float()
                ref = ref_logits[step]
                worst = max(worst, float((got - ref).abs().max() / ref.abs().max()))
            assert worst < 2e-2, worst
        assert model.config._attn_implementation == impl_before
        T = ids.shape[-1] + n_new
        assert fc.qcache._k.lens == [T] * len(kv) and cache.get_seq_length() == T
        assert torch.equal(fc.qcache._k.q[0, :, :, :T], qc._k.q[0, :, :, :T])
        assert torch.equal(fc.qcache._k.scales[0, :T], qc._k.scales[0, :T])
        assert fc.estimated_bytes() == qc.estimated_bytes()
        assert fc.qcache._k.stage is None and fc.qcache._v.stage is None  # no fp16 copy was ever made

    bench = KVCacheBenchmarker(model, tok, device="cuda")
    a = bench.generate_with_quantized_kv("<150>", n_new, mode=mode)
    bench.fused_attention = True
    b = bench.generate_with_quantized_kv("<150>", n_new, mode=mode)
    assert b[1] == n_new and b[2] == a[2]
    res = bench.benchmark_method(["<60>", "<61>"], f"quant_{mode}", max_new_tokens=6)
    assert res["total_new_tokens"] == 12 and res["est_kv_cache_mb_avg"] > 0


def test_llama_grouped_query_model_all_paths():
    """A Llama-architecture model (random init; 8 query heads on 2 kv heads, head_dim 128 — Llama-3-8B's
    grouping): the tuple, staged and fused-attention decodes all run through HF's Llama attention; the
    fused path takes the MFMA kernel. Teacher-forced logits of the fused path track the tuple path."""
    import efficient_llm_inference_amd as E
    from efficient_llm_inference_amd import KVCacheBenchmarker
    from efficient_llm_inference_amd.benchmarking import from_legacy_tuple, to_legacy_tuple
    from efficient_llm_inference_amd.benchmarking.offline import load_model
    from efficient_llm_inference_amd.quantization import fused_attention as FA
    model, tok = load_model("llama-mini", "cuda", torch.float16)
    bench = KVCacheBenchmarker(model, tok, device="cuda")
    n_new = 16
    bench.inplace_decode = False
    a = bench.generate_with_quantized_kv("<200>", n_new, mode="mixed")
    bench.inplace_decode = True
    b = bench.generate_with_quantized_kv("<200>", n_new, mode="mixed")
    assert a == b and a[1] == n_new  # staged == tuple, token for token
    for method in ("sliding_window", "chunked_cache", "paged_attention", "budget_cache"):
        res = bench.benchmark_method(["<150>"], method, max_new_tokens=4, window_size=64, chunk_size=16, keep_last=32,
                                     block_size=16, old_budget=8)
        assert res["total_new_tokens"] == 4, method
    with torch.no_grad():
        ids = tok("<200>", return_tensors="pt").input_ids.cuda()
        out = model(input_ids=ids, use_cache=True)
        kv = to_legacy_tuple(out.past_key_values)
        assert kv[0][0].shape[1] == 2 and kv[0][0].shape[-1] == 128  # kv heads, head_dim
        qc = E.QuantizedKVCache(len(kv), "mixed", incremental=False)
        qc.init_from_prompt_past(kv)
        logits = out.logits[:, -1, :]
        ref_logits, toks = [], []
        for _ in range(n_new):
            nxt = torch.argmax(logits, dim=-1, keepdim=True)
            toks.append(nxt)
            out = model(input_ids=nxt, use_cache=True, past_key_values=from_legacy_tuple(qc.to_past_key_values()))
            logits = out.logits[:, -1, :]
            qc.append_from_past(to_legacy_tuple(out.past_key_values))
            ref_logits.append(logits.float())
        fc = FA.FusedQuantizedCache(len(kv), mode="mixed", reserve=ids.shape[-1] + n_new)
        with FA.fused_attention(model, fc) as cache:
            model(input_ids=ids, use_cache=True, past_key_values=cache)
            worst = 0.0
            for step, nxt in enumerate(toks):
                got = model(input_ids=nxt, use_cache=True, past_key_values=cache).logits[:, -1, :].float()
                worst = max(worst, float((got - ref_logits[step]).abs().max() / ref_logits[step].abs().max()))
        assert worst < 2e-2, worst
        assert torch.equal(fc.qcache._k.q[0, :, :, :ids.shape[-1]], qc._k.q[0, :, :, :ids.shape[-1]])
        assert fc.estimated_bytes() == qc.estimated_bytes()
    bench.fused_attention = True
    c = bench.generate_with_quantized_kv("<200>", n_new, mode="mixed")
    assert c[1] == n_new and c[2] == a[2]


def test_fused_attention_bf16_model():
    """bf16 weights (the Llama-family dtype): the fused path quantises bf16 K/V (scales rounded to
    bf16, like the reference's) and attends from bf16 queries; same cache size as the staged path and
    mostly the same tokens (bf16 logits of a random-init model have near-ties, so not all)."""
    from efficient_llm_inference_amd import KVCacheBenchmarker
    from efficient_llm_inference_amd.benchmarking.offline import load_model
    model, tok = load_model("gpt2-mini", "cuda", torch.bfloat16)
    b = KVCacheBenchmarker(model, tok, device="cuda")
    staged = b.generate_with_quantized_kv("<90>", 12, "mixed")
    b.fused_attention = True
    fused = b.generate_with_quantized_kv("<90>", 12, "mixed")
    assert fused[1] == 12 and fused[2] == staged[2]


def test_fused_attention_rejects_unsupported_head_dim(rig):
    """gpt2-tiny has head_dim 16: the fused kernel refuses it loudly (no silent fallback)."""
    from efficient_llm_inference_amd._lib import KvqError
    bench = rig[0]
    bench.fused_attention = True
    try:
        with pytest.raises(KvqError):
            bench.generate_with_quantized_kv("<20>", 4, mode="int8")
    finally:
        bench.fused_attention = False
    assert bench.model.config._attn_implementation != "kvq_fused"


def test_sliding_window_nll_through_hip_equals_plain_slicing(rig):
    """N4: compute_sliding_window_nll trims its cache through kvq_window_compact; the trim is an exact
    copy, so the NLL equals the reference's loop (src/evaluation/quality.py:60-121) with plain slicing
    on the same model — bit for bit. A window longer than the text equals no trimming at all."""
    import math

    from efficient_llm_inference_amd.evaluation import compute_sliding_window_nll
    _, model, tok, from_legacy, to_legacy = rig
    text, W = "<40>", 7
    got = compute_sliding_window_nll(model, tok, text, window_size=W, device="cuda")
    ids = tok(text, return_tensors="pt").input_ids.cuda()
    nll_sum, n_tok, past, prev = 0.0, 0, None, ids[:, :1]
    with torch.no_grad():
        for i in range(1, ids.size(1)):
            out = model(input_ids=prev, use_cache=True, past_key_values=past)
            logits = out.logits[:, -1, :]
            kv = tuple((k[:, :, -W:, :].contiguous(), v[:, :, -W:, :].contiguous()) if k.size(2) > W else (k, v)
                       for k, v in to_legacy(out.past_key_values))
            past = from_legacy(kv)
            target = ids[:, i]
            nll_sum += -torch.log_softmax(logits.float(), dim=-1).gather(1, target.unsqueeze(1)).item()
            n_tok += 1
            prev = target.unsqueeze(1)
    want = nll_sum / n_tok
    assert got == (want, math.exp(want))
    wide = compute_sliding_window_nll(model, tok, text, window_size=4096, device="cuda")
    assert wide[0] > 0 and wide != got


def test_config2_gpt2_quant_int8_prompt512_new512():
    """BASELINE configs[1] at its stated size: gpt2 architecture (12 layers x 12 heads x 64, random-init
    weights: there is no network for the checkpoint), quant_int8, prompt 512 + 512 new tokens through
    generate_with_quantized_kv (reference src/benchmarking/benchmarker.py:422-491). The in-place staged
    path and the reference-shaped tuple path (dequantise everything, rebuild the cache, cat) emit the same
    512 tokens, and both report est_mb == the oracle's estimated_bytes for 1024 stored tokens."""
    from efficient_llm_inference_amd import KVCacheBenchmarker
    from efficient_llm_inference_amd.benchmarking.offline import load_model
    model, tok = load_model("gpt2", "cuda", torch.float16)
    bench = KVCacheBenchmarker(model, tok, device="cuda")
    assert tok("<512>", return_tensors="pt").input_ids.shape[1] == 512
    bench.inplace_decode = True
    a = bench.generate_with_quantized_kv("<512>", 512, mode="int8")
    bench.inplace_decode = False
    b = bench.generate_with_quantized_kv("<512>", 512, mode="int8")
    assert a[1] == b[1] == 512
    assert a[0] == b[0], "in-place and tuple decode paths diverged"
    want_mb = O.estimated_bytes("int8", 12, 1, 12, 1024, 64, 2) / 2**20  # fp16 stored scales: itemsize 2
    assert a[2] == b[2] == want_mb
    res = bench.benchmark_method(["<512>"], method="quant_int8", max_new_tokens=512)
    assert res["total_new_tokens"] == 512 and res["est_kv_cache_mb_avg"] == want_mb and res["tokens_per_sec"] > 0


@pytest.mark.parametrize("arch,mode", [("gpt2-mini", "int8"), ("gpt2-mini", "mixed"), ("llama-mini", "mixed")])
def test_graph_decode_equals_eager_fused_decode(arch, mode):
    """KVCacheBenchmarker.graph_decode: the decode step captured into a HIP graph (model forward +
    kvq_decode_step_dev per layer + arg-max) and replayed per token emits the same tokens as the eager
    fused-attention loop, leaves the same quantised store behind (layer 0 compared bit for bit through
    estimated bytes and a final attend) and reports the same cache size."""
    from efficient_llm_inference_amd import KVCacheBenchmarker
    from efficient_llm_inference_amd.benchmarking.offline import load_model
    model, tok = load_model(arch, "cuda", torch.float16)
    bench = KVCacheBenchmarker(model, tok, device="cuda")
    bench.fused_attention = True
    eager = bench.generate_with_quantized_kv("<40>", 24, mode=mode)
    bench.graph_decode = True
    graphed = bench.generate_with_quantized_kv("<40>", 24, mode=mode)
    again = bench.generate_with_quantized_kv("<40>", 24, mode=mode)  # a second capture on the same model
    assert graphed[1] == eager[1] == 24 and graphed[2] == eager[2]
    assert graphed[0] == eager[0] == again[0]
    short = bench.generate_with_quantized_kv("<40>", 2, mode=mode)  # fewer steps than the eager prologue
    bench.graph_decode = False
    assert short == bench.generate_with_quantized_kv("<40>", 2, mode=mode)
    bench.fused_attention = False
    bench.graph_decode = True
    with pytest.raises(RuntimeError):
        bench.generate_with_quantized_kv("<40>", 4, mode=mode)


G9_METHODS = ["sliding_window", "quant_int8", "quant_int4", "quant_mixed", "paged_attention", "chunked_cache",
              "prefix_window", "strided_cache", "block_cache", "budget_cache"]


@pytest.fixture(scope="module")
def g9_rig():
    """the offline gpt2-tiny in fp32 on the GPU: the reference run of g9_benchmarker.npz was CPU fp32, and the byte
    counts it reports (scale itemsize = the KV dtype's, ops.py:271-290; tensor bytes of the chunked cache) follow the
    KV dtype"""
    from efficient_llm_inference_amd import KVCacheBenchmarker
    from efficient_llm_inference_amd.benchmarking.offline import load_model
    from tests.util import CallRecorder
    model, tok = load_model("gpt2-tiny", "cuda", torch.float32)
    rec = CallRecorder(model)
    return KVCacheBenchmarker(rec, tok, device="cuda"), rec


@pytest.mark.parametrize("method", G9_METHODS)
def test_decode_loops_equal_the_reference_run(g9_rig, method):
    """Part B of g9_benchmarker.npz: the reference's quant_* / sliding / chunked / paged / sparse loops (run by
    tests/golden/make_golden.py behind a generator-side DynamicCache.from_legacy_cache / to_legacy_cache adapter)
    against this package's, float-independent observables only: dict keys + order, value kinds, token count, the
    estimated cache MB, the policy parameters echoed back, and the cache length the model saw at every forward."""
    from tests.conftest import load_golden
    from tests.util import value_kinds
    g = load_golden("g9_benchmarker.npz")
    bm, rec = g9_rig
    kw = {k: int(v) for k, v in (str(s).split("=") for s in g["B.kwargs"])}
    prompts = [str(p) for p in g["prompts"]]
    rec.calls.clear()
    res = bm.benchmark_method(prompts, method=method, **kw)
    assert list(res.keys()) == [str(k) for k in g[f"B.{method}.keys"]]
    want = str(g[f"B.{method}.kinds"][0])
    got = value_kinds(res)
    assert got[:5] + got[6:] == want[:5] + want[6:], (got, want)  # [5] = gpu_peak_mb: None on the reference's CPU run
    assert res["total_new_tokens"] == int(g[f"B.{method}.total_new_tokens"][0])
    ref_mb = float(g[f"B.{method}.est_kv_cache_mb_avg"][0])
    if method == "paged_attention":
        # the reference picks the block dtype by DEVICE STRING (benchmarker.py:521: fp16 on "cuda", fp32 on "cpu"): the
        # fixture's CPU run counted 4-byte elements, this GPU run counts 2-byte ones — same blocks, half the bytes
        ref_mb *= 0.5
    assert (np.isnan(ref_mb) and np.isnan(res["est_kv_cache_mb_avg"])) or abs(res["est_kv_cache_mb_avg"] - ref_mb) <= 1e-12 * ref_mb, (res["est_kv_cache_mb_avg"], ref_mb)
    for k2 in ("window_size", "block_size", "chunk_size", "prefix_len", "stride", "keep_per_block", "old_budget"):
        assert (-1 if res[k2] is None else res[k2]) == int(g[f"B.{method}.{k2}"][0]), k2
    assert rec.calls == [tuple(c) for c in g[f"B.{method}.calls"].tolist()], method


# fp32 model on the card against the reference's CPU fp32 run. The eviction loops move fp32 rows unchanged: GEMM / softmax
# summation order is all that differs (measured 3e-7). The quantised and the paged loops pick their working dtype by DEVICE
# STRING in the reference (benchmarker.py:452 compute_dtype, :521 block dtype: fp16 on "cuda", fp32 on "cpu"), and this
# package does the same: on the card the dequantised values / the blocks are rounded to fp16 where the fixture's CPU run
# kept fp32 — measured 1.5e-5 – 2.0e-5 on logits of magnitude 0.93. The reference's own loops are at least 3.4e-4 apart from one another on these logits (INT4 / mixed; INT8 /
# INT4 8.5e-3; tests/test_benchmarker_cpu.py::test_g9_part_c_tells_the_methods_apart), so either bound tells them apart.
G9_LOGIT_TOL = 5e-5
G9_LOGIT_TOL_EXACT_ROWS = 2e-6


@pytest.mark.parametrize("method", G9_METHODS)
def test_decode_loop_logits_equal_the_reference_run(g9_rig, method):
    """Part C of g9_benchmarker.npz (round 4): the token fed and the next-token logits of EVERY forward of the reference's
    loops (prefill + 12 decode steps x 3 prompts), against this package's loops with the same fp32 model on the GPU.
    Independent of `_emulate` (a9's logits were pinned by the builder's reading only): same tokens at every
    step, logits within G9_LOGIT_TOL absolute (G9_LOGIT_TOL_EXACT_ROWS for the loops that only move rows)."""
    from tests.conftest import load_golden
    g = load_golden("g9_benchmarker.npz")
    bm, rec = g9_rig
    kw = {k: int(v) for k, v in (str(s).split("=") for s in g["B.kwargs"])}
    rec.clear()
    bm.benchmark_method([str(p) for p in g["prompts"]], method=method, **kw)
    ref_fed, ref_logits = g[f"C.{method}.fed"], g[f"C.{method}.logits"]
    assert rec.fed == ref_fed.tolist(), method  # every greedy choice of the reference run, reproduced
    got = np.stack(rec.logits)
    assert got.shape == ref_logits.shape
    err = float(np.abs(got - ref_logits).max())
    print(f"{method}: max |logit - reference run| = {err:.3e} over {got.shape[0]} forwards (max |logit| {float(np.abs(ref_logits).max()):.3f})")
    tol = G9_LOGIT_TOL if method.startswith("quant_") or method == "paged_attention" else G9_LOGIT_TOL_EXACT_ROWS
    assert err <= tol, (method, err, tol)


def test_generate_functions_equal_the_reference_run(g9_rig):
    from tests.conftest import load_golden
    g = load_golden("g9_benchmarker.npz")
    bm, _ = g9_rig
    for i, p in enumerate(str(p) for p in g["prompts"]):
        for mode in ("int8", "int4", "mixed"):
            _, n_new, est_mb = bm.generate_with_quantized_kv(p, 12, mode=mode)
            assert [n_new, est_mb] == g[f"B.generate_with_quantized_kv.{mode}.{i}"].tolist(), (mode, i)
        _, n_new, est_mb = bm.generate_with_chunked_cache(p, 12, chunk_size=4, keep_last=6)
        assert [n_new, est_mb] == g[f"B.generate_with_chunked_cache.{i}"].tolist(), i
        _, n_new, alloc_mb, used_mb, nblocks = bm.generate_with_paged_attention(p, 12, block_size=4)
        r_new, r_alloc, r_used, r_blocks = g[f"B.generate_with_paged_attention.{i}"].tolist()
        assert [n_new, alloc_mb, used_mb, nblocks] == [r_new, r_alloc / 2, r_used / 2, r_blocks], i  # fp16 blocks on "cuda" (reference :521)
