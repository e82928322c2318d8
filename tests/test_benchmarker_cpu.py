"""Host-logic tests of the benchmarker, the cache-format shim and the sharding helpers (CPU).
BASELINE config 1: gpt2-family full_cache on CPU (plumbing, no GPU)."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import ROOT

from tests.conftest import load_golden
from tests.util import CallRecorder, value_kinds

# the dict the REFERENCE's benchmark_method returned when tests/golden/make_golden.py ran it (reference
# benchmarker.py:811-832): key order from the reference run, not from reading
G9 = load_golden("g9_benchmarker.npz")
REF_KEYS = [str(k) for k in G9["A.full_cache.keys"]]


@pytest.fixture(scope="module")
def bench_cpu():
    from efficient_llm_inference_amd import KVCacheBenchmarker
    from efficient_llm_inference_amd.benchmarking.offline import load_model
    model, tok = load_model("gpt2-tiny", "cpu", torch.float32)
    return KVCacheBenchmarker(model, tok, device="cpu")


def test_full_cache_plumbing_on_cpu(bench_cpu):
    res = bench_cpu.benchmark_method(["The quick brown fox", "<40>"], method="full_cache", max_new_tokens=64)
    assert list(res.keys()) == REF_KEYS
    assert res["method"] == "full_cache" and res["total_new_tokens"] == 128
    assert res["tokens_per_sec"] > 0 and math.isnan(res["est_kv_cache_mb_avg"]) and res["gpu_peak_mb"] is None
    assert res["window_size"] is None and res["chunk_size"] is None


@pytest.mark.parametrize("method", ["no_cache", "full_cache"])
def test_benchmark_method_equals_the_reference_run(method):
    """Part A of g9_benchmarker.npz: the reference's own KVCacheBenchmarker.benchmark_method, run unmodified on the
    offline gpt2-tiny (CPU fp32), against this package's on the same model: same keys in the same order, same value
    kinds (None / NaN / float / int / str), same token count, same forwards (tokens fed, cache length seen) in the same
    order."""
    from efficient_llm_inference_amd import KVCacheBenchmarker
    from efficient_llm_inference_amd.benchmarking.offline import load_model
    model, tok = load_model("gpt2-tiny", "cpu", torch.float32)
    rec = CallRecorder(model)
    bm = KVCacheBenchmarker(rec, tok, device="cpu")
    prompts = [str(p) for p in G9["prompts"]]
    res = bm.benchmark_method(prompts, method=method, max_new_tokens=int(G9["A.max_new_tokens"][0]))
    assert list(res.keys()) == [str(k) for k in G9[f"A.{method}.keys"]]
    assert value_kinds(res) == str(G9[f"A.{method}.kinds"][0])
    assert res["total_new_tokens"] == int(G9[f"A.{method}.total_new_tokens"][0])
    assert rec.calls == [tuple(c) for c in G9[f"A.{method}.calls"].tolist()]


def test_generate_functions_equal_the_reference_run(bench_cpu):
    """generate_with_cache / generate_no_cache: the text (hence every greedy token) and n_new the reference produced"""
    for i, p in enumerate(str(p) for p in G9["prompts"]):
        for name, fn in (("with_cache", bench_cpu.generate_with_cache), ("no_cache", bench_cpu.generate_no_cache)):
            text, n_new = fn(p, int(G9["A.max_new_tokens"][0]))
            assert n_new == int(G9[f"A.generate_{name}.{i}.n_new"][0])
            assert text.encode("utf-8", "surrogatepass") == G9[f"A.generate_{name}.{i}.text_utf8"].tobytes(), (name, i)


def test_cache_equals_no_cache_tokens(bench_cpu):
    """greedy decoding with and without the KV cache yields the same text (fp32, CPU)."""
    t1, n1 = bench_cpu.generate_with_cache("<12>", 10)
    t2, n2 = bench_cpu.generate_no_cache("<12>", 10)
    assert n1 == 10 and (n2 == 10 or n2 < 10)  # no_cache may stop at EOS (reference :97-98)
    if n2 == 10:
        assert t1 == t2


def test_method_validation(bench_cpu):
    with pytest.raises(AssertionError, match="Invalid method"):
        bench_cpu.benchmark_method(["x"], method="quant_int2")
    for m in ("quant_int8", "quant_int4", "quant_mixed", "sliding_window", "chunked_cache", "paged_attention",
              "prefix_window", "strided_cache", "block_cache", "budget_cache"):
        with pytest.raises(RuntimeError, match="MI355X"):  # hot path has no CPU implementation
            bench_cpu.benchmark_method(["<40>"], method=m, max_new_tokens=2, window_size=8, keep_last=8, chunk_size=4,
                                       prefix_len=2, block_size=8, keep_per_block=2, old_budget=4)
    # fused attention over the quantised store: same rule, and the model's attention setting is restored
    before = bench_cpu.model.config._attn_implementation
    bench_cpu.fused_attention = True
    try:
        with pytest.raises(RuntimeError, match="MI355X"):
            bench_cpu.benchmark_method(["<40>"], method="quant_mixed", max_new_tokens=2)
    finally:
        bench_cpu.fused_attention = False
    assert bench_cpu.model.config._attn_implementation == before


def test_cache_format_shim_roundtrip():
    from efficient_llm_inference_amd.benchmarking import from_legacy_tuple, to_legacy_tuple
    tup = tuple((torch.randn(1, 2, 5, 4), torch.randn(1, 2, 5, 4)) for _ in range(3))
    cache = from_legacy_tuple(tup)
    back = to_legacy_tuple(cache)
    assert len(back) == 3 and all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(tup, back))
    assert to_legacy_tuple(tup) == tup


def test_config_surface():
    from efficient_llm_inference_amd import BenchmarkConfig, CacheConfig, Config, QuantizationConfig
    c = Config(device="cpu")
    assert (c.model_name, c.seed, c.max_new_tokens, c.batch_size) == ("gpt2", 42, 64, 1)
    a = torch.rand(3)
    Config(device="cpu")  # re-seeds: same stream again (reference config.py:32-37)
    b = torch.rand(3)
    assert torch.equal(a, b)
    assert (QuantizationConfig().mode, QuantizationConfig().eps) == ("int8", 1e-8)
    cc = CacheConfig()
    assert (cc.window_size, cc.block_size, cc.chunk_size, cc.keep_last) == (256, 64, 64, 256)
    assert BenchmarkConfig().window_sizes == [64, 128, 256, 512]


def test_sharding_helpers_single_process():
    from efficient_llm_inference_amd import sharding
    prompts = [f"p{i}" for i in range(10)]
    parts = [sharding.shard_prompts(prompts, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == sorted(prompts) and [len(p) for p in parts] == [3, 3, 2, 2]
    rows = [sharding.shard_batch_rows(64, r, 8) for r in range(8)]
    assert all(len(r) == 8 for r in rows) and rows[3][0] == 24
    rows = [sharding.shard_batch_rows(10, r, 4) for r in range(4)]
    assert [list(r) for r in rows] == [[0, 1, 2], [3, 4, 5], [6, 7], [8, 9]]
    assert sharding.aggregate_results({"total_new_tokens": 5, "elapsed_sec": 1.0})["n_ranks"] == 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from efficient_llm_inference_amd import KVCacheBenchmarker, sharding
        from efficient_llm_inference_amd.benchmarking.offline import load_model
        model, tok = load_model("gpt2-tiny", "cpu", torch.float32)
        b = KVCacheBenchmarker(model, tok, device="cpu")
        prompts = ["<8>", "<9>", "<10>", "<11>", "<12>"]
        res = sharding.benchmark_sharded(b, prompts, "full_cache", max_new_tokens=4)
        sharding.barrier()
        slowest = sharding.max_over_ranks(1.0 + rank)  # bench.py's timing reduction
        rows = list(sharding.shard_batch_rows(5))
        # local_mode: inside it this rank is a single process — no collective is touched (only rank 1 enters the block
        # here: a barrier or reduction inside it would hang the pair), outside it the group is back
        if rank == 1:
            with sharding.local_mode():
                assert sharding.world() == (0, 1) and sharding.backend() is None
                sharding.barrier()
                assert sharding.max_over_ranks(7.0) == 7.0
                loc = sharding.benchmark_sharded(b, prompts, "full_cache", max_new_tokens=2)
                assert loc["n_ranks"] == 1 and loc["total_new_tokens"] == 10
        assert sharding.world() == (rank, world_size)
        assert sharding.max_over_ranks(float(rank)) == 1.0
        q.put((rank, res["total_new_tokens"], res["n_prompts"], res["n_ranks"], res["elapsed_sec"], res["tokens_per_sec"],
               slowest, rows))
    finally:
        dist.destroy_process_group()


def test_sharded_benchmark_world_size_2_gloo():
    """N>1 path: prompts sharded over 2 ranks, counters aggregated by one all_reduce pair."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, total, n_prompts, n_ranks, elapsed, tps, slowest, rows in out:
        assert total == 20 and n_prompts == 5 and n_ranks == 2  # 5 prompts x 4 tokens, both ranks agree
        assert abs(tps - total / elapsed) < 1e-9
        assert slowest == 2.0  # MAX over ranks of (1 + rank)
        assert rows == ([0, 1, 2] if rank == 0 else [3, 4])
    assert out[0][4] == out[1][4]  # max-over-ranks elapsed identical on every rank


def test_quality_helpers(bench_cpu):
    from efficient_llm_inference_amd._lib import KvqError
    from efficient_llm_inference_amd.evaluation import (compute_perplexity, compute_sliding_window_nll,
                                                        text_similarity, token_agreement_rate)
    assert text_similarity("abc", "abc") == 1.0 and text_similarity("abc", "xyz") == 0.0
    assert token_agreement_rate([1, 2, 3, 4], [1, 9, 3]) == 2 / 3 and token_agreement_rate([], [1]) == 0.0
    nll, ppl = compute_perplexity(bench_cpu.model, bench_cpu.tokenizer, ["<24>", "hello there"], device="cpu")
    assert nll > 0 and abs(ppl - __import__("math").exp(nll)) < 1e-9
    # the sliding-window NLL trims through the HIP path: on host tensors it raises like the rest of the package
    with pytest.raises(KvqError):
        compute_sliding_window_nll(bench_cpu.model, bench_cpu.tokenizer, "<20>", window_size=4, device="cpu")


def test_quality_helpers_match_reference_goldens():
    """N4: text_similarity / token_agreement_rate pinned to the reference's own outputs
    (tests/golden/g8_round2.npz, generated by importing src/evaluation/quality.py:124-150)."""
    from efficient_llm_inference_amd.evaluation import text_similarity, token_agreement_rate
    from tests.conftest import load_golden
    g8 = load_golden("g8_round2.npz")
    for a, b, r in zip(g8["textsim.a"].tolist(), g8["textsim.b"].tolist(), g8["textsim.ratio"].tolist()):
        assert text_similarity(a, b) == r, (a, b)
    for a, b, (la, lb), r in zip(g8["tokagree.a"].tolist(), g8["tokagree.b"].tolist(), g8["tokagree.len"].tolist(),
                                 g8["tokagree.rate"].tolist()):
        assert token_agreement_rate(a[:la], b[:lb]) == r, (a[:la], b[:lb])


def test_fused_attention_rejects_unsupported_variants():
    """sliding-window / soft-capping / attention-sink arguments must stop the fused attention
    function, not be swallowed by **kwargs (the guard runs before any tensor is touched)."""
    import types

    from efficient_llm_inference_amd.quantization import fused_attention as FA
    plain = types.SimpleNamespace(layer_idx=0)
    FA._reject_unsupported_variants(plain, {"sliding_window": None, "is_causal": True})  # nothing to reject
    for kw in ({"sliding_window": 4096}, {"softcap": 50.0}, {"s_aux": object()}):
        with pytest.raises(RuntimeError, match=next(iter(kw))):
            FA._reject_unsupported_variants(plain, kw)
    for attr in ("sliding_window", "sinks", "attn_logit_softcapping"):
        with pytest.raises(RuntimeError, match=attr):
            FA._reject_unsupported_variants(types.SimpleNamespace(layer_idx=0, **{attr: 7}), {})
    if FA.available():  # through the registered function itself
        prev = FA._ACTIVE
        FA._ACTIVE = object()
        try:
            with pytest.raises(RuntimeError, match="sliding_window"):
                FA._fused_attention_forward(plain, None, None, None, sliding_window=128)
        finally:
            FA._ACTIVE = prev


def test_g9_part_c_tells_the_methods_apart():
    """Part C of the fixture (the reference's per-forward logits, tests/golden/make_golden.py::gen_benchmarker) is only a pin
    if different loops give different numbers: every pair of methods is at least 5 x the GPU test's tolerance apart
    (tests/test_gpu_benchmarker.py::G9_LOGIT_TOL), and each method's logits are finite with the shape of its call list."""
    methods = [k.split(".")[1] for k in G9.files if k.startswith("C.") and k.endswith(".logits")]
    assert len(methods) == 10
    for m in methods:
        lg = G9[f"C.{m}.logits"]
        assert lg.dtype == np.float32 and np.isfinite(lg).all() and lg.shape == (len(G9[f"B.{m}.calls"]), 260)
        assert len(G9[f"C.{m}.fed"]) == lg.shape[0]
    for i, a in enumerate(methods):
        for b in methods[i + 1:]:
            d = float(np.abs(G9[f"C.{a}.logits"] - G9[f"C.{b}.logits"]).max())
            assert d > 1e-4, (a, b, d)  # closest pair: quant_int4 / quant_mixed, 3.4e-4
