#!/usr/bin/env python3
"""Generate golden vectors for the KV quantize / dequantize / eviction path.

Run ONLY in the build container, where the reference checkout is mounted read-only:

    python tests/golden/make_golden.py [/root/reference [out_dir]]      (out_dir: default = this directory)

It imports the reference's own Python functions (CPU branches: src/quantization/ops.py:88-90
and :120-133 are taken because no GPU is visible) and stores inputs + outputs as ``.npz``
(plain arrays, loadable with ``allow_pickle=False``).  The fixtures are DATA; no reference
source text is stored.  The reference never travels to the GPU box — these files do.

Groups:
  G1/G4  a1-a4 on slice shapes, fp32 + fp16 (+bf16), N(0,1) and heavy-tailed
  G2     rounding / packing known-answer test
  G3     all-zero slice
  G5     QuantizedKVCache end-to-end (init_from_prompt_past, append_from_past,
         to_past_key_values, estimated_bytes) for int8 / int4 / mixed
  G6     trim_kv_sliding_window  (T<W, T==W, T>W)
  G7     chunk_summarize_kv      (old%chunk==0, !=0, T<=keep_last, keep_last=0, re-application)
  G9     g9_benchmarker.npz (round 3): the REFERENCE's KVCacheBenchmarker run on the build's offline gpt2-tiny
         (random init, seed 42, byte tokenizer; CPU fp32). Part A (adapter-free; the reference's no_cache / full_cache
         paths run unmodified on the installed transformers): benchmark_method dict key order, value kinds,
         total_new_tokens, generated token ids. Part B (the quant_* / sliding / chunked / paged / sparse loops call
         DynamicCache.from_legacy_cache / to_legacy_cache, which transformers >= 5 removed): the same reference code
         run with a GENERATOR-SIDE adapter that gives DynamicCache those two methods back; only float-independent
         observables are kept (n_new, est_mb, paged alloc / used MB and block counts, the cache length the model saw
         at every forward, dict key order / value kinds). Part C (round 4): from the same part-B runs, the token fed
         and the fp32 next-token logits of EVERY forward of every method — the reference's quantised / evicting decode
         loops as numbers, for a tolerance-level comparison with this package's loops on the GPU (fp32 model; GEMM
         summation order differs between the CPU run and the card, so not bit-level).
  G8     round-2 additions: window_size == 0 edge of the trims (the reference's ``-0:`` slice keeps
         everything), budget-policy index lists at more lengths, and known-answer tests of the
         quality helpers text_similarity / token_agreement_rate (src/evaluation/quality.py:124-150)
"""
import os
import sys

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, REF)

from src.cache.implementations import (  # noqa: E402
    PagedKVCache,
    chunk_summarize_kv,
    trim_kv_block_old,
    trim_kv_budget_old,
    trim_kv_prefix_window,
    trim_kv_sliding_window,
    trim_kv_strided,
)
from src.quantization.ops import (  # noqa: E402
    QuantizedKVCache,
    dequantize_int4_per_tensor_packed,
    dequantize_int8_per_tensor,
    quantize_int4_per_tensor_packed,
    quantize_int8_per_tensor,
)

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else HERE
os.makedirs(OUT, exist_ok=True)
TD = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def to_np(t: torch.Tensor) -> np.ndarray:
    """bf16 has no numpy dtype: store its bit pattern as uint16."""
    t = t.detach().contiguous()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    return t.numpy().copy()


def make_input(shape, dtype, dist, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    if dist == "heavy":
        m = torch.rand(shape, generator=g) < 0.01
        x = torch.where(m, x * 10.0, x)
    elif dist == "tiny":
        x = x * 1e-6
    return x.to(TD[dtype])


def gen_slices():
    out = {}
    shapes = {"gpt2": (1, 12, 1, 64), "gpt2m": (1, 16, 1, 64), "llama": (1, 8, 1, 128),
              "odd": (2, 3, 1, 5), "one": (1, 1, 1, 1), "b2": (2, 4, 1, 16)}
    case = 0
    for sname, shape in shapes.items():
        for dtype in ("f32", "f16", "bf16"):
            for dist in ("normal", "heavy", "tiny"):
                x = make_input(shape, dtype, dist, 1000 + case)
                key = f"{sname}.{dtype}.{dist}"
                q8, s8 = quantize_int8_per_tensor(x)
                p4, s4, last = quantize_int4_per_tensor_packed(x)
                out[key + ".x"] = to_np(x)
                out[key + ".q8"] = to_np(q8)
                out[key + ".s8"] = to_np(s8.reshape(1))
                out[key + ".p4"] = to_np(p4)
                out[key + ".s4"] = to_np(s4.reshape(1))
                out[key + ".last"] = np.array([last], dtype=np.int64)
                for od in ("f16", "f32", "bf16"):
                    out[key + f".dq8.{od}"] = to_np(dequantize_int8_per_tensor(q8, s8, TD[od]))
                    out[key + f".dq4.{od}"] = to_np(dequantize_int4_per_tensor_packed(p4, s4, last, TD[od]))
                case += 1
    np.savez_compressed(os.path.join(OUT, "g1_slices.npz"), **out)
    print("g1_slices:", len(out), "arrays")


def gen_kat():
    out = {}
    x = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, 7.0, -7.0, 3.5])
    p4, s4, last = quantize_int4_per_tensor_packed(x)
    out["kat4.x"] = to_np(x)
    out["kat4.p4"] = to_np(p4)
    out["kat4.s4"] = to_np(s4.reshape(1))
    out["kat4.dq.f16"] = to_np(dequantize_int4_per_tensor_packed(p4, s4, last, torch.float16))
    # int8 half-to-even KAT: scale is exactly 1.0 when max|x| = 127
    x8 = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 126.5, 127.0, -127.0, 63.5, 0.49999, 100.5])
    q8, s8 = quantize_int8_per_tensor(x8)
    out["kat8.x"] = to_np(x8)
    out["kat8.q8"] = to_np(q8)
    out["kat8.s8"] = to_np(s8.reshape(1))
    # zero slices (G3)
    for dtype in ("f32", "f16", "bf16"):
        z = torch.zeros(1, 4, 1, 8, dtype=TD[dtype])
        q8, s8 = quantize_int8_per_tensor(z)
        p4, s4, last = quantize_int4_per_tensor_packed(z)
        out[f"zero.{dtype}.q8"] = to_np(q8)
        out[f"zero.{dtype}.s8"] = to_np(s8.reshape(1))
        out[f"zero.{dtype}.p4"] = to_np(p4)
        out[f"zero.{dtype}.s4"] = to_np(s4.reshape(1))
        out[f"zero.{dtype}.dq8.f16"] = to_np(dequantize_int8_per_tensor(q8, s8, torch.float16))
        out[f"zero.{dtype}.dq4.f16"] = to_np(dequantize_int4_per_tensor_packed(p4, s4, last, torch.float16))
    np.savez_compressed(os.path.join(OUT, "g2_kat.npz"), **out)
    print("g2_kat:", len(out), "arrays")


def gen_cache():
    """G5: the container end-to-end.  Input KV is [L,2,B,H,T,D]; one extra token is appended
    through append_from_past (ops.py:323-330, which reads k[:,:,-1:,:])."""
    out = {}
    cfgs = {"tiny": (2, 1, 2, 5, 8), "gpt2ish": (3, 1, 12, 9, 64), "llamaish": (2, 1, 8, 6, 128),
            "odd": (2, 2, 3, 4, 5)}
    case = 0
    for cname, (L, B, H, T, D) in cfgs.items():
        for dtype in ("f32", "f16"):
            kv = make_input((L, 2, B, H, T + 1, D), dtype, "heavy" if case % 2 else "normal", 2000 + case)
            out[f"{cname}.{dtype}.kv"] = to_np(kv)
            for mode in ("int8", "int4", "mixed"):
                qc = QuantizedKVCache(n_layers=L, mode=mode, device="cpu", compute_dtype=TD[dtype])
                prompt = tuple((kv[l, 0, :, :, :T], kv[l, 1, :, :, :T]) for l in range(L))
                qc.init_from_prompt_past(prompt)
                full = tuple((kv[l, 0], kv[l, 1]) for l in range(L))
                qc.append_from_past(full)
                past = qc.to_past_key_values()
                deq = torch.stack([torch.stack([k, v]) for k, v in past])  # [L,2,B,H,T+1,D]
                key = f"{cname}.{dtype}.{mode}"
                out[key + ".deq"] = to_np(deq)
                out[key + ".bytes"] = np.array([qc.estimated_bytes()], dtype=np.int64)
                # stored scales per layer per token (k then v), in the storage dtype
                ks = torch.stack([torch.stack(layer.k_scales) for layer in qc.layers])
                vs = torch.stack([torch.stack(layer.v_scales) for layer in qc.layers])
                out[key + ".scales"] = to_np(torch.stack([ks, vs], dim=1))  # [L,2,T+1]
                kq = torch.stack([torch.cat(layer.k_store, dim=2) for layer in qc.layers])
                vq = torch.stack([torch.cat(layer.v_store, dim=2) for layer in qc.layers])
                out[key + ".kq"] = to_np(kq)
                out[key + ".vq"] = to_np(vq)
            case += 1
    np.savez_compressed(os.path.join(OUT, "g5_cache.npz"), **out)
    print("g5_cache:", len(out), "arrays")


def gen_cache_bf16():
    """bf16 KV (what Llama-family models produce) with fp16 compute_dtype (what the reference's
    decode loop asks for on a GPU, benchmarker.py:452): scales are stored rounded to bf16."""
    out = {}
    for cname, (L, B, H, T, D) in {"llamaish": (2, 1, 8, 6, 128), "odd": (2, 2, 3, 4, 5)}.items():
        kv = make_input((L, 2, B, H, T + 1, D), "bf16", "heavy", 2500 + L + D)
        out[f"{cname}.kv"] = to_np(kv)
        for mode in ("int8", "int4", "mixed"):
            qc = QuantizedKVCache(n_layers=L, mode=mode, device="cpu", compute_dtype=torch.float16)
            qc.init_from_prompt_past(tuple((kv[l, 0, :, :, :T], kv[l, 1, :, :, :T]) for l in range(L)))
            qc.append_from_past(tuple((kv[l, 0], kv[l, 1]) for l in range(L)))
            past = qc.to_past_key_values()
            key = f"{cname}.{mode}"
            out[key + ".deq"] = to_np(torch.stack([torch.stack([k, v]) for k, v in past]))
            out[key + ".bytes"] = np.array([qc.estimated_bytes()], dtype=np.int64)
            ks = torch.stack([torch.stack(layer.k_scales) for layer in qc.layers])
            vs = torch.stack([torch.stack(layer.v_scales) for layer in qc.layers])
            out[key + ".scales"] = to_np(torch.stack([ks, vs], dim=1))
    np.savez_compressed(os.path.join(OUT, "g5b_cache_bf16.npz"), **out)
    print("g5b_cache_bf16:", len(out), "arrays")


def gen_evict():
    out = {}
    case = 0
    # G6 sliding window
    for dtype in ("f16", "f32"):
        for (T, W) in ((5, 8), (8, 8), (13, 8), (40, 1)):
            x = make_input((1, 3, T, 16), dtype, "normal", 3000 + case)
            (k, v), = trim_kv_sliding_window(((x, x * 2),), W)
            out[f"win.{dtype}.T{T}.W{W}.x"] = to_np(x)
            out[f"win.{dtype}.T{T}.W{W}.k"] = to_np(k)
            out[f"win.{dtype}.T{T}.W{W}.v"] = to_np(v)
            case += 1
    # G7 chunk summary
    for dtype in ("f16", "f32", "bf16"):
        for (T, chunk, keep) in ((40, 8, 8), (45, 8, 8), (6, 8, 8), (33, 4, 0), (300, 64, 16), (19, 64, 3)):
            x = make_input((2, 3, T, 16), dtype, "heavy", 4000 + case)
            (k, v), = chunk_summarize_kv(((x, -x),), chunk_size=chunk, keep_last=keep)
            out[f"chunk.{dtype}.T{T}.c{chunk}.k{keep}.x"] = to_np(x)
            out[f"chunk.{dtype}.T{T}.c{chunk}.k{keep}.k"] = to_np(k)
            case += 1
    # re-application trajectory of generate_with_chunked_cache (benchmarker.py:610-626):
    # each decode step appends one token then re-summarises the already summarised cache.
    T, lens = 32768, []
    x = torch.zeros(1, 1, T, 2)
    for _ in range(4):
        (x, _), = chunk_summarize_kv(((x, x),), chunk_size=64, keep_last=256)
        lens.append(x.size(2))
        x = torch.cat([x, torch.zeros(1, 1, 1, 2)], dim=2)
    out["chunk.trajectory"] = np.array(lens, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "g6_evict.npz"), **out)
    print("g6_evict:", len(out), "arrays; trajectory", lens)


def gen_sparse():
    """N3: index-select eviction family + PagedKVCache. Token rows are made identifiable
    (value = token index + small per-channel offset) so the kept index list can be read back."""
    out = {}
    cases = []
    for T in (5, 40, 41, 100, 300, 1000):
        x = (torch.arange(T, dtype=torch.float32)[None, None, :, None] + torch.arange(8)[None, None, None, :] / 16.0)
        x = x.expand(1, 2, T, 8).contiguous().half()
        for (W, P) in ((8, 0), (8, 4), (32, 3), (256, 32)):
            (k, _), = trim_kv_prefix_window(((x, x),), prefix_len=P, window_size=W)
            out[f"prefix.T{T}.W{W}.P{P}"] = to_np(k[0, 0, :, 0].float().round().long())
            for stride in (1, 3, 4):
                (k, _), = trim_kv_strided(((x, x),), window_size=W, stride=stride, prefix_len=P)
                out[f"strided.T{T}.W{W}.P{P}.s{stride}"] = to_np(k[0, 0, :, 0].float().round().long())
            for (bs, kpb) in ((16, 4), (64, 8), (7, 7)):
                (k, _), = trim_kv_block_old(((x, x),), window_size=W, block_size=bs, keep_per_block=kpb, prefix_len=P)
                out[f"block.T{T}.W{W}.P{P}.b{bs}.k{kpb}"] = to_np(k[0, 0, :, 0].float().round().long())
            for budget in (0, 1, 5, 64):
                (k, _), = trim_kv_budget_old(((x, x),), window_size=W, old_budget=budget, prefix_len=P)
                out[f"budget.T{T}.W{W}.P{P}.n{budget}"] = to_np(k[0, 0, :, 0].float().round().long())
    # paged cache: append 21 tokens with block_size 8 -> 3 blocks, stitched back
    g = torch.Generator().manual_seed(5000)
    kv = torch.randn(2, 2, 3, 21, 8, generator=g).half()
    pc = PagedKVCache(block_size=8, device="cpu", dtype=torch.float16)
    for t in range(21):
        pc.append(kv[0, :, :, t:t + 1], kv[1, :, :, t:t + 1])
    k, v = pc.get_kv()
    out["paged.kv"] = to_np(kv)
    out["paged.k"] = to_np(k)
    out["paged.v"] = to_np(v)
    out["paged.meta"] = np.array([pc.num_blocks(), pc.allocated_bytes(), pc.used_bytes()], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "g7_sparse.npz"), **out)
    print("g7_sparse:", len(out), "arrays")


def gen_round2():
    """G8 (g8_round2.npz). Strings travel as fixed-width unicode arrays (plain data, no pickle)."""
    from src.evaluation.quality import text_similarity, token_agreement_rate
    out = {}
    # window_size == 0: `k[:, :, -0:, :]` is the whole tensor (implementations.py:137-139, :151-152)
    for T in (1, 7):
        x = (torch.arange(T, dtype=torch.float32)[None, None, :, None] + torch.arange(8)[None, None, None, :] / 16.0)
        x = x.expand(1, 2, T, 8).contiguous().half()
        (k, _), = trim_kv_sliding_window(((x, x),), 0)
        out[f"win0.T{T}"] = to_np(k[0, 0, :, 0].float().round().long())
        for P in (0, 3):
            (k, _), = trim_kv_prefix_window(((x, x),), prefix_len=P, window_size=0)
            out[f"prefix0.T{T}.P{P}"] = to_np(k[0, 0, :, 0].float().round().long())
    # budget policy (fp32 linspace truncated to integers, :279-282) at lengths where the spacing is not integral
    for T in (97, 513, 2049, 4096, 16385, 32768):
        x = (torch.arange(T, dtype=torch.float32)[None, None, :, None]).expand(1, 1, T, 2).contiguous()
        for (W, P, n) in ((8, 0, 7), (256, 32, 64), (33, 5, 100), (1, 1, 3)):
            (k, _), = trim_kv_budget_old(((x, x),), window_size=W, old_budget=n, prefix_len=P)
            out[f"budget.T{T}.W{W}.P{P}.n{n}"] = to_np(k[0, 0, :, 0].round().long())
    # quality helpers
    pairs = [("", ""), ("abc", "abc"), ("abc", "xyz"), ("the quick brown fox", "the quick brown dog"),
             ("KV cache quantisation", "KV-cache quantization"), ("aaaa", "aa"), ("hello world", "world hello"),
             ("a", ""), ("Once upon a time, there was a model.", "Once upon a time there was a small model"),
             ("0123456789" * 5, "0123456789" * 4 + "abcdefghij")]
    out["textsim.a"] = np.array([a for a, _ in pairs])
    out["textsim.b"] = np.array([b for _, b in pairs])
    out["textsim.ratio"] = np.array([text_similarity(a, b) for a, b in pairs], dtype=np.float64)
    g = torch.Generator().manual_seed(8000)
    toks = []
    for i, (la, lb) in enumerate(((0, 0), (0, 5), (5, 0), (1, 1), (8, 8), (8, 5), (5, 8), (64, 64), (64, 100), (3, 3))):
        a = torch.randint(0, 4, (la,), generator=g).tolist()
        b = torch.randint(0, 4, (lb,), generator=g).tolist()
        if i == 9:
            b = list(a)
        toks.append((a, b))
    width = max(max(len(a), len(b)) for a, b in toks)
    pad = lambda t: t + [-1] * (width - len(t))  # noqa: E731
    out["tokagree.a"] = np.array([pad(a) for a, _ in toks], dtype=np.int64)
    out["tokagree.b"] = np.array([pad(b) for _, b in toks], dtype=np.int64)
    out["tokagree.len"] = np.array([[len(a), len(b)] for a, b in toks], dtype=np.int64)
    out["tokagree.rate"] = np.array([token_agreement_rate(a, b) for a, b in toks], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "g8_round2.npz"), **out)
    print("g8_round2:", len(out), "arrays")


def gen_benchmarker():
    """G9 (g9_benchmarker.npz): see the module docstring. Needs the repo root on sys.path for the OFFLINE MODEL only
    (efficient_llm_inference_amd.benchmarking.offline: a random-init HF GPT2LMHeadModel + byte tokenizer — the
    reference's examples download 'gpt2', which this container cannot); every benchmarker line that runs is the
    reference's."""
    import math
    repo = os.path.dirname(os.path.dirname(HERE))
    if repo not in sys.path:
        sys.path.insert(1, repo)
    import transformers
    from efficient_llm_inference_amd.benchmarking.offline import load_model
    from src.benchmarking.benchmarker import KVCacheBenchmarker as RefBM
    model, tok = load_model("gpt2-tiny", "cpu", torch.float32)

    class Recorder:
        """the model as the reference calls it, noting (new tokens fed, cache length seen) per forward"""
        def __init__(self, m):
            self.m, self.config, self.calls, self.fed, self.logits = m, m.config, [], [], []
        def __call__(self, **kw):
            past = kw.get("past_key_values")
            plen = 0
            if past is not None:
                plen = int(past.get_seq_length()) if hasattr(past, "get_seq_length") else int(past[0][0].size(2))
            self.calls.append((int(kw["input_ids"].shape[-1]), plen))
            res = self.m(**kw)
            # part C: the last token fed and the next-token logits of every forward (one prompt at a time: batch 1)
            self.fed.append(int(kw["input_ids"][0, -1]))
            self.logits.append(res.logits[0, -1, :].detach().to(torch.float32).numpy().copy())
            return res
        def clear(self):
            self.calls.clear()
            self.fed.clear()
            self.logits.clear()

    def kinds(d):  # value kind per key: n None, a NaN, f float, i int, s str
        out = []
        for v in d.values():
            out.append("n" if v is None else "s" if isinstance(v, str) else "i" if isinstance(v, (int, np.integer)) and not isinstance(v, bool)
                       else "a" if isinstance(v, float) and math.isnan(v) else "f")
        return "".join(out)

    prompts = ["The quick brown fox", "<23>", "<31>"]
    out = {"prompts": np.array(prompts), "model": np.array(["gpt2-tiny seed 42 fp32 cpu (benchmarking/offline.py)"]),
           "transformers": np.array([transformers.__version__])}
    rec = Recorder(model)
    ref = RefBM(rec, tok, device="cpu")
    ids = lambda text: np.frombuffer(text.encode("utf-8", "surrogatepass"), dtype=np.uint8).copy()  # noqa: E731

    # ---- part A: adapter-free ------------------------------------------------------------------------------------
    NEW_A = 16
    for method in ("no_cache", "full_cache"):
        rec.calls.clear()
        res = ref.benchmark_method(prompts, method=method, max_new_tokens=NEW_A)
        out[f"A.{method}.keys"] = np.array(list(res.keys()))
        out[f"A.{method}.kinds"] = np.array([kinds(res)])
        out[f"A.{method}.total_new_tokens"] = np.array([res["total_new_tokens"]], dtype=np.int64)
        out[f"A.{method}.calls"] = np.array(rec.calls, dtype=np.int64)
    out["A.max_new_tokens"] = np.array([NEW_A], dtype=np.int64)
    for i, p in enumerate(prompts):
        for name, fn in (("with_cache", ref.generate_with_cache), ("no_cache", ref.generate_no_cache)):
            text, n_new = fn(p, NEW_A)
            out[f"A.generate_{name}.{i}.text_utf8"] = ids(text)
            out[f"A.generate_{name}.{i}.n_new"] = np.array([n_new], dtype=np.int64)

    # ---- part B: the reference's loops behind a generator-side DynamicCache adapter --------------------------------
    DC = transformers.DynamicCache
    added = []
    if not hasattr(DC, "from_legacy_cache"):
        DC.from_legacy_cache = classmethod(lambda cls, past: cls(ddp_cache_data=past))
        added.append("from_legacy_cache")
    if not hasattr(DC, "to_legacy_cache"):
        DC.to_legacy_cache = lambda self: tuple((layer.keys, layer.values) for layer in self.layers)
        added.append("to_legacy_cache")
    out["B.adapter"] = np.array(["generator-side: DynamicCache." + ", DynamicCache.".join(added) if added else "none needed"])
    kw = dict(max_new_tokens=12, window_size=8, block_size=4, chunk_size=4, keep_last=6, prefix_len=2, stride=2, keep_per_block=1, old_budget=3)
    out["B.kwargs"] = np.array([f"{k}={v}" for k, v in kw.items()])
    try:
        for method in ("sliding_window", "quant_int8", "quant_int4", "quant_mixed", "paged_attention", "chunked_cache",
                       "prefix_window", "strided_cache", "block_cache", "budget_cache"):
            rec.clear()
            res = ref.benchmark_method(prompts, method=method, **kw)
            out[f"C.{method}.fed"] = np.array(rec.fed, dtype=np.int64)
            out[f"C.{method}.logits"] = np.stack(rec.logits).astype(np.float32)
            out[f"B.{method}.keys"] = np.array(list(res.keys()))
            out[f"B.{method}.kinds"] = np.array([kinds(res)])
            out[f"B.{method}.total_new_tokens"] = np.array([res["total_new_tokens"]], dtype=np.int64)
            out[f"B.{method}.est_kv_cache_mb_avg"] = np.array([res["est_kv_cache_mb_avg"]], dtype=np.float64)
            out[f"B.{method}.calls"] = np.array(rec.calls, dtype=np.int64)
            for k2 in ("window_size", "block_size", "chunk_size", "prefix_len", "stride", "keep_per_block", "old_budget"):
                out[f"B.{method}.{k2}"] = np.array([-1 if res[k2] is None else res[k2]], dtype=np.int64)
        for i, p in enumerate(prompts):
            for mode in ("int8", "int4", "mixed"):
                _, n_new, est_mb = ref.generate_with_quantized_kv(p, 12, mode=mode)
                out[f"B.generate_with_quantized_kv.{mode}.{i}"] = np.array([n_new, est_mb], dtype=np.float64)
            _, n_new, est_mb = ref.generate_with_chunked_cache(p, 12, chunk_size=4, keep_last=6)
            out[f"B.generate_with_chunked_cache.{i}"] = np.array([n_new, est_mb], dtype=np.float64)
            _, n_new, alloc_mb, used_mb, nblocks = ref.generate_with_paged_attention(p, 12, block_size=4)
            out[f"B.generate_with_paged_attention.{i}"] = np.array([n_new, alloc_mb, used_mb, nblocks], dtype=np.float64)
    finally:
        for name in added:
            delattr(DC, name)
    np.savez_compressed(os.path.join(OUT, "g9_benchmarker.npz"), **out)
    print("g9_benchmarker:", len(out), "arrays; adapter:", out["B.adapter"][0])


if __name__ == "__main__":
    torch.manual_seed(42)
    gen_slices()
    gen_kat()
    gen_cache()
    gen_cache_bf16()
    gen_evict()
    gen_sparse()
    gen_round2()
    gen_benchmarker()
