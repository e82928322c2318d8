#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X KV-cache dequantise hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload llama3_8b_mixed_seq16k]

Workload (BASELINE.json configs[3], the one the target is quoted on): Llama-3-8B KV shape
[L=32, 2, B=1, H_kv=8, T=16384, D=128], ``quant_mixed`` = INT8 keys + packed-INT4 values,
synthetic N(0,1) fp16 KV (seed 42) quantised once by the HIP quantise kernels before timing.

One STEP = one full ``QuantizedKVCache.to_past_key_values()`` of that cache — what the
reference does before every decode forward (reference src/quantization/ops.py:345-355,
src/benchmarking/benchmarker.py:470): dequantise all 32 layers of K (INT8 -> fp16) and V
(INT4 -> fp16): exactly two kernel launches, inputs resident in HBM.

value  = algorithmic bytes of the step (INT8 3.0 B/elt, INT4 2.5 B/elt) x ranks
         / max-over-ranks wall time, GB/s.  Weak scaling: every rank holds its own prompt's
         cache (batch shard, no data-path collective).
roofline = the INT4 dequantise kernel (north-star kernel): algorithmic bytes per launch
         (1,342,177,280) / its mean duration, timed with HIP events on the launch stream
         inside the timed region; peak 8000 GB/s (MI355X HBM3E spec).
cpu_baseline = the C restatement of the reference's algorithm (oracle/kvq_oracle.c, scalar,
         1 core) on a bounded sample of the same INT4 workload, on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: (L, B, H, T, D, mode)
    "llama3_8b_mixed_seq16k": (32, 1, 8, 16384, 128, "mixed"),
    "gpt2m_int4_seq4k": (24, 1, 16, 4096, 64, "int4"),
    "gpt2_int8_seq1k": (12, 1, 12, 1024, 64, "int8"),
}
# BASELINE.json configs[1]/[2] as decode loops through KVCacheBenchmarker (random-init weights of
# the named architecture; see benchmarking/offline.py): --workload decode:<arch>:<method>
DECODE_DEFAULT = ("gpt2", "quant_int8", 512, 512)  # arch, method, prompt tokens, new tokens
# BASELINE.json configs[4]: Llama-3-8B sliding_window + chunk_summary, seq 32K, batch 64 sharded
# over 8 GPUs = 8 batch rows per GPU: per-rank KV [L=32, 2, B=8, H=8, T=32768, D=128] fp16 = 32 GiB
EVICT = {"llama3_8b_evict_seq32k": (32, 8, 8, 32768, 128, 256, 64, 256)}  # L,B,H,T,D,window,chunk,keep_last
# scope row N1 (second form): one decode step's attention over the quantised store, all layers
ATTN = {  # name: (L, B, Hq, Hkv, T, D, mode)
    "llama3_8b_decode_attn_seq16k": (32, 1, 32, 8, 16384, 128, "mixed"),
    "llama3_8b_decode_attn_seq16k_b8": (32, 8, 32, 8, 16384, 128, "mixed"),
    "gpt2_decode_attn_seq1k": (12, 1, 12, 12, 1024, 64, "int8"),
    "llama2_7b_decode_attn_seq4k_b8": (32, 8, 32, 32, 4096, 128, "mixed"),  # multi-head (one query head per kv head)
    "llama32_1b_decode_attn_seq16k_b8": (16, 8, 32, 8, 16384, 64, "mixed"),  # grouped-query at head_dim 64
}
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
BYTES_PER_ELT = {"int8": 3.0, "int4": 2.5}  # q read + fp16 write


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="llama3_8b_mixed_seq16k",
                    help="one of %s, or decode:<arch>:<method>[:<prompt_tokens>:<new_tokens>] "
                         "(e.g. decode:gpt2:quant_int8:512:512; steps = prompts per rank)" % sorted(list(WORKLOADS) + list(ATTN) + list(EVICT)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rotate-caches", type=int, default=2,
                    help="independent quantised caches visited round-robin by consecutive steps, so that no "
                         "step re-reads lines the 256 MiB Infinity Cache may still hold (1 = the decode loop's "
                         "behaviour: the same cache every step; its 256 MiB INT4 store then stays cache-resident)")
    ap.add_argument("--cpu-sample-layers", type=int, default=32)
    ap.add_argument("--tunable", action="append", default=[], metavar="KEY=VALUE",
                    help="A-B only: kvq_set_tunable(KEY, VALUE) before the run (include/kvq_hip.h lists the keys)")
    ap.add_argument("--fused-attention", action="store_true",
                    help="decode:* workloads with quant_* methods: attend straight over the quantised store "
                         "(kvq_decode_attn) instead of the staged fp16 copy")
    return ap.parse_args()


def cpu_baseline(L, B, H, T, D, sample_layers):
    """Time the scalar C port of the reference's INT4 dequantise on `sample_layers` layers of
    the same V set (host buffers, one thread)."""
    import numpy as np
    from oracle import c_oracle as C
    n_layers = max(1, min(sample_layers, L))
    rng = np.random.default_rng(42)
    q = rng.integers(0, 256, size=(n_layers, B, H, T, D // 2), dtype=np.uint8)
    sc = (rng.random((n_layers, T), dtype=np.float32) * 0.02 + 0.001).astype(np.float32)
    C.dequantize_tokens(q[:1], sc[:1], "int4", D, "f16")  # warm (page in, load lib)
    dt = float("inf")
    for _ in range(3):  # best of three passes over the sample (~10 s of CPU work at the default)
        t0 = time.perf_counter()
        C.dequantize_tokens(q, sc, "int4", D, "f16")
        dt = min(dt, time.perf_counter() - t0)
    n = n_layers * B * H * T * D
    # the quantise side of the same path (rows a1/a2): scalar C port, fp16 -> INT4, on 4 layers
    nq_layers = max(1, min(4, n_layers))
    xq = (rng.standard_normal((nq_layers, B, H, T, D), dtype=np.float32)).astype(np.float16)
    C.quantize_tokens(xq[:1, :, :, :64], "int4")
    tq0 = time.perf_counter()
    C.quantize_tokens(xq, "int4")
    dq_s = time.perf_counter() - tq0
    nq = nq_layers * B * H * T * D
    # the reference's literal call structure (per-slice op chains + T-way cat) on a small sample
    import torch as _t
    from oracle import literal_loop as LL
    Ts = min(T, 2048)
    qs = [_t.from_numpy(q[0, :, :, t:t + 1, :].copy()) for t in range(Ts)]
    ss = [_t.tensor(float(sc[0, t]), dtype=_t.float16) for t in range(Ts)]
    _t.set_num_threads(os.cpu_count() or 1)
    LL.dequantize_slices(qs[:64], ss[:64], "int4", D, _t.float16)
    t1 = time.perf_counter()
    LL.dequantize_slices(qs, ss, "int4", D, _t.float16)
    dl = time.perf_counter() - t1
    n_lit = B * H * Ts * D
    return {
        "value": round(n * BYTES_PER_ELT["int4"] / dt / 1e9, 4),
        "unit": "GB/s",
        "cores": 1,
        "kind": "port",
        "sample": f"INT4->fp16 dequantise of {n_layers}/{L} layers of the V set "
                  f"[{n_layers},{B},{H},{T},{D}] ({n} elements, best of 3 passes: {dt:.2f} s), oracle/kvq_oracle.c scalar",
        "host_cores_available": os.cpu_count(),
        "quantise": {"value": round(nq * BYTES_PER_ELT["int4"] / dq_s / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                     "sample": f"fp16->INT4 per-token quantise of {nq_layers}/{L} layers [{nq_layers},{B},{H},{T},{D}] "
                               f"({dq_s:.2f} s), oracle/kvq_oracle.c scalar"},
        "literal_loop": {"value": round(n_lit * BYTES_PER_ELT["int4"] / dl / 1e9, 5), "unit": "GB/s",
                         "sample": f"per-slice dequantise + {Ts}-way cat of one layer's V [{B},{H},{Ts},{D}] with torch-CPU ops "
                                   f"({dl:.2f} s), the reference's own call structure (oracle/literal_loop.py)",
                         "torch_threads": _t.get_num_threads()},
    }


def run_decode(args, rank, world, dev):
    """BASELINE configs[1]: decode tokens/sec + KV-cache MB through KVCacheBenchmarker.benchmark_method
    (reference benchmarker.py:643-832), prompts sharded over ranks, counters aggregated once."""
    import efficient_llm_inference_amd as E
    from efficient_llm_inference_amd import sharding
    from efficient_llm_inference_amd.benchmarking.offline import load_model
    parts = args.workload.split(":")
    arch = parts[1] if len(parts) > 1 else DECODE_DEFAULT[0]
    method = parts[2] if len(parts) > 2 else DECODE_DEFAULT[1]
    n_prompt = int(parts[3]) if len(parts) > 3 else DECODE_DEFAULT[2]
    n_new = int(parts[4]) if len(parts) > 4 else DECODE_DEFAULT[3]
    model, tok = load_model(arch, "cuda", torch.float16)
    bm = E.KVCacheBenchmarker(model, tok, device="cuda")
    bm.fused_attention = bool(args.fused_attention)
    prompts = [f"<{n_prompt}>"] * (args.steps * world)
    for _ in range(args.warmup):
        bm.benchmark_method([f"<{min(n_prompt, 64)}>"], method=method, max_new_tokens=8)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    res = sharding.benchmark_sharded(bm, prompts, method, max_new_tokens=n_new)
    base = sharding.benchmark_sharded(bm, prompts, "full_cache", max_new_tokens=n_new)
    if rank == 0:
        cfg = model.config
        print(json.dumps({
            "metric": "decode tokens/sec + KV-cache MB", "value": round(res["tokens_per_sec"], 2), "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(res["elapsed_sec"] / max(1, args.steps) * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16 model, u8/u4 KV", "data": "synthetic",
            "config": {"workload": args.workload, "arch": arch, "method": method, "fused_attention": bool(args.fused_attention),
                       "prompt_tokens": n_prompt,
                       "new_tokens": n_new, "weights": "random-init (offline)",
                       "layers": getattr(cfg, "num_hidden_layers", None) or cfg.n_layer,
                       "heads": getattr(cfg, "num_attention_heads", None) or cfg.n_head,
                       "kv_heads": getattr(cfg, "num_key_value_heads", None) or getattr(cfg, "num_attention_heads", None) or cfg.n_head,
                       "head_dim": getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads,
                       "parallelism": f"prompt-shard x{world}, one all_reduce of counters"},
            "est_kv_cache_mb": round(res["est_kv_cache_mb_avg"], 3),
            "full_cache_tokens_per_sec": round(base["tokens_per_sec"], 2),
            "vs_full_cache": round(res["tokens_per_sec"] / base["tokens_per_sec"], 3),
            "gpu_peak_mb": res["gpu_peak_mb"],
        }), flush=True)


def run_evict(args, rank, world, dev):
    """configs[4] per-rank slice: one STEP = trim_kv_sliding_window + chunk_summarize_kv over the
    whole legacy tuple (64 tensors of [8,8,32768,128] fp16): two launches, inputs resident."""
    import efficient_llm_inference_amd as E
    from efficient_llm_inference_amd import sharding
    L, B, H, T, D, W, chunk, keep = EVICT[args.workload]
    torch.manual_seed(42 + rank)
    past = tuple((torch.randn(B, H, T, D, device=dev, dtype=torch.float16),
                  torch.randn(B, H, T, D, device=dev, dtype=torch.float16)) for _ in range(L))
    from efficient_llm_inference_amd.kernels import chunk_summary_len
    Tout = chunk_summary_len(T, chunk, keep)
    n_t = 2 * L
    bytes_win = 4.0 * n_t * B * H * W * D                      # 2 B read + 2 B write per kept element
    bytes_pool = 2.0 * n_t * B * H * D * (T + Tout)            # read every token once, write Tout rows
    step_bytes = bytes_win + bytes_pool

    def step(evs=None):
        if evs is not None:
            evs[0].record()
        E.trim_kv_sliding_window(past, W)
        if evs is not None:
            evs[1].record()
        E.chunk_summarize_kv(past, chunk_size=chunk, keep_last=keep)
        if evs is not None:
            evs[2].record()

    for _ in range(args.warmup):
        step()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events[i])
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    w_ms = sum(e[0].elapsed_time(e[1]) for e in events) / args.steps
    p_ms = sum(e[1].elapsed_time(e[2]) for e in events) / args.steps
    elapsed = sharding.max_over_ranks(elapsed, dev)  # the step takes as long as the slowest rank
    if rank == 0:
        print(json.dumps({
            "metric": "KV eviction GB/s vs HBM roofline (sliding_window + chunk_summary step)",
            "value": round(step_bytes * world / (elapsed / args.steps) / 1e9, 1), "unit": "GB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "dtype_detail": "fp16 in/out, fp32 accumulate",
            "data": "synthetic",
            "config": {"workload": args.workload, "shape_per_rank_L2BHTD": [L, 2, B, H, T, D], "window": W,
                       "chunk_size": chunk, "keep_last": keep, "global_batch": B * world,
                       "parallelism": f"batch-shard x{world} (8 rows per GPU), no collective"},
            "roofline": {"kernel": "chunk_pool_vec_k<f16>", "bound": "hbm",
                         "achieved": round(bytes_pool / (p_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(bytes_pool / (p_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": int(bytes_pool), "avg_launch_ms": round(p_ms, 4)},
            "roofline_window": {"kernel": "copy_rows_k", "achieved": round(bytes_win / (w_ms * 1e-3) / 1e9, 1),
                                "frac": round(bytes_win / (w_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                "algorithmic_bytes_per_launch": int(bytes_win), "avg_launch_ms": round(w_ms, 4),
                                "note": "includes torch.empty + host launch path; 0.5 GB per launch"},
        }), flush=True)


def _traffic(workload, key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), or None"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(workload, {}).get(key)
    except Exception:
        return None


def run_attn(args, rank, world, dev):
    """One STEP = the attention of one decode step over the quantised store of every layer
    (kvq_decode_attn per layer: split-T partial kernel + merge), new token's exact K/V included.
    Side measurement: the same step as the staged path runs it (torch SDPA over the fp16 copy)."""
    import efficient_llm_inference_amd as E
    from efficient_llm_inference_amd import kernels as K
    from efficient_llm_inference_amd import sharding
    L, B, Hq, Hkv, T, D, mode = ATTN[args.workload]
    kk, vk = {"int8": ("int8", "int8"), "int4": ("int4", "int4"), "mixed": ("int8", "int4")}[mode]
    torch.manual_seed(42 + rank)
    qc = E.QuantizedKVCache(n_layers=L, mode=mode, device="cuda", compute_dtype=torch.float16)
    qc.reserve(T)
    for g0 in range(0, L, 4):  # quantise 4 layers at a time: bounded fp16 scratch
        n = min(4, L - g0)
        qc._k.append([torch.randn(B, Hkv, T, D, device=dev, dtype=torch.float16) for _ in range(n)], g0=g0)
        qc._v.append([torch.randn(B, Hkv, T, D, device=dev, dtype=torch.float16) for _ in range(n)], g0=g0)
    q = torch.randn(L, B, Hq, D, device=dev, dtype=torch.float16)
    kn = torch.randn(L, B, Hkv, D, device=dev, dtype=torch.float16)
    vn = torch.randn(L, B, Hkv, D, device=dev, dtype=torch.float16)
    out = torch.empty(L, B, Hq, D, device=dev, dtype=torch.float16)
    ws = torch.empty(K.decode_attn_workspace(B, Hq, Hkv, T, D), device=dev, dtype=torch.float32)
    sm = D ** -0.5
    layer_bytes = B * Hkv * T * (K.packed_dim(kk, D) + K.packed_dim(vk, D)) + 8 * T  # rows + two fp32 scales per token
    step_bytes = L * layer_bytes

    def step():
        for i in range(L):
            K.decode_attn(q[i], qc._k.q[i], qc._k.scales[i], kk, qc._v.q[i], qc._v.scales[i], vk, T, out[i], ws, sm,
                          kn[i], vn[i])

    for _ in range(args.warmup):
        step()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for _ in range(args.steps):
        step()
    ev[1].record()
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    layer_ms = ev[0].elapsed_time(ev[1]) / (args.steps * L)
    elapsed = sharding.max_over_ranks(elapsed, dev)

    # the staged path's attention on the same shapes: SDPA over an fp16 copy of ONE layer
    kf = torch.randn(B, Hkv, T + 1, D, device=dev, dtype=torch.float16)
    vf = torch.randn(B, Hkv, T + 1, D, device=dev, dtype=torch.float16)
    q4 = q[0].unsqueeze(2)
    sd = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(12):
        if it == 2:
            sd[0].record()
        torch.nn.functional.scaled_dot_product_attention(q4, kf, vf, scale=sm, enable_gqa=Hq != Hkv)
    sd[1].record()
    torch.cuda.synchronize()
    sdpa_ms = sd[0].elapsed_time(sd[1]) / 10
    if rank == 0:
        print(json.dumps({
            "metric": f"decode attention over the quantised KV store, GB/s vs HBM roofline ({kk.upper()} K + {vk.upper()} V)",
            "value": round(step_bytes * world / (elapsed / args.steps) / 1e9, 1), "unit": "GB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "dtype_detail": "int8 / packed-int4 store, fp16 query, fp32 accumulate, fp16 out", "data": "synthetic",
            "config": {"workload": args.workload, "shape_L_B_Hq_Hkv_T_D": [L, B, Hq, Hkv, T, D], "mode": mode,
                       "step": "one decode step: kvq_decode_attn per layer (2 launches each), host launch gaps included",
                       "bytes_per_step": int(step_bytes), "parallelism": f"batch-shard x{world}, no collective"},
            "roofline": {"kernel": ("decode_attn_partial_mfma_k" if D in (64, 128) and 3 <= Hq // Hkv <= 16 else "decode_attn_partial_k (or _mfma_k under --tunable attn_mfma_min_nq)") + " + decode_attn_merge_k (per layer call)", "bound": "hbm",
                         "achieved": round(layer_bytes / (layer_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(layer_bytes / (layer_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": _traffic(args.workload, "decode_attn_per_layer_call"),
                         "algorithmic_bytes_per_launch": int(layer_bytes), "avg_launch_ms": round(layer_ms, 5),
                         "timer": "HIP events around the timed region / (steps * layers)"},
            "staged_path_sdpa": {"what": "torch SDPA over an fp16 copy of one layer's KV (what the staged decode runs)",
                                 "avg_ms": round(sdpa_ms, 5), "fp16_bytes": int(4 * B * Hkv * (T + 1) * D),
                                 "speedup_of_fused": round(sdpa_ms / layer_ms, 3)},
            "est_kv_cache_mb": round(qc.estimated_bytes() / 2**20, 3),
            "fp16_kv_cache_mb": round(L * 4 * B * Hkv * T * D / 2**20, 3),
        }), flush=True)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(1, n_dev)  # ranks > GPUs (a rehearsal on a 1-GPU box) share the card
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = None
    if world > 1:
        # RCCL over xGMI for the one scalar reduction of a run; the data path itself needs no
        # collective. If RCCL cannot come up (IPC / driver trouble) the same reduction runs over
        # gloo so that the per-rank measurements are still reported.
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            backend = "nccl"
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)  # surfaces RCCL initialisation errors here, not inside the timed region
            torch.cuda.synchronize()
        except Exception as exc:  # noqa: BLE001
            print(f"[bench] rank {rank}: RCCL unavailable ({exc!r}); using gloo for the timing reduction", file=sys.stderr)
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group("gloo", rank=rank, world_size=world)
            backend = "gloo"

    import efficient_llm_inference_amd as E
    from efficient_llm_inference_amd import _lib, sharding

    _lib.load()
    for kv in args.tunable:
        key, _, val = kv.partition("=")
        _lib.set_tunable(key, int(val))
    if args.workload.startswith("decode"):
        run_decode(args, rank, world, dev)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.workload in ATTN:
        run_attn(args, rank, world, dev)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.workload in EVICT:
        run_evict(args, rank, world, dev)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.workload not in WORKLOADS:
        raise SystemExit(f"unknown workload {args.workload}")
    L, B, H, T, D, mode = WORKLOADS[args.workload]
    kk, vk = {"int8": ("int8", "int8"), "int4": ("int4", "int4"), "mixed": ("int8", "int4")}[mode]

    # ---- build the quantised cache once (each rank its own prompt: seed 42 + rank) ------------
    torch.manual_seed(42 + rank)
    qc = E.QuantizedKVCache(n_layers=L, mode=mode, device="cuda", compute_dtype=torch.float16)
    qc.reserve(T)
    past = []
    for _ in range(L):  # the legacy tuple layout: 2L separately allocated [B,H,T,D] tensors
        past.append((torch.randn(B, H, T, D, device=dev, dtype=torch.float16),
                     torch.randn(B, H, T, D, device=dev, dtype=torch.float16)))
    qc.init_from_prompt_past(tuple(past))
    caches = [qc]
    for _ in range(max(1, args.rotate_caches) - 1):  # same content, different HBM lines
        extra = E.QuantizedKVCache(n_layers=L, mode=mode, device="cuda", compute_dtype=torch.float16)
        extra.reserve(T)
        extra.init_from_prompt_past(tuple(past))
        caches.append(extra)
    torch.cuda.synchronize()
    est_mb = qc.estimated_bytes() / 2**20

    # side measurement, outside the timed region: the prefill quantise kernels (rows a1/a2) on the
    # same tensors, re-quantising into the same store (identical bytes every time)
    from efficient_llm_inference_amd import kernels as _k
    quant_info = {}
    for name, store, tensors in (("k", qc._k, [k for k, _ in past]), ("v", qc._v, [v for _, v in past])):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ws = store._workspace(L * T)
        for it in range(6):
            if it == 1:
                ev[0].record()
            _k.quant_tokens(tensors, store.q[:, :, :, :T], store.scales[:, :T], ws, store.kind)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 5
        qbytes = L * B * H * T * D * BYTES_PER_ELT[store.kind]  # 2 B read + 1 or 0.5 B written per element
        quant_info[f"quant_{store.kind}"] = {"avg_launch_ms": round(ms, 4), "achieved": round(qbytes / (ms * 1e-3) / 1e9, 1),
                                             "frac": round(qbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                             "algorithmic_bytes_per_launch": int(qbytes), "set": name.upper()}
    del past
    torch.cuda.empty_cache()

    n_elts = L * B * H * T * D  # per K or V set
    bytes_k = n_elts * BYTES_PER_ELT[kk]
    bytes_v = n_elts * BYTES_PER_ELT[vk]
    step_bytes = bytes_k + bytes_v

    # two rotating output sets so consecutive steps never write the same lines
    outs = [(torch.empty(L, B, H, T, D, device=dev, dtype=torch.float16),
             torch.empty(L, B, H, T, D, device=dev, dtype=torch.float16)) for _ in range(2)]

    def step(i, evs=None):
        ko, vo = outs[i & 1]
        c = caches[i % len(caches)]
        if evs is not None:
            evs[0].record()
        c._k.dequant(torch.float16, out=ko)  # all layers of K: one launch
        if evs is not None:
            evs[1].record()
        c._v.dequant(torch.float16, out=vo)  # all layers of V: one launch
        if evs is not None:
            evs[2].record()

    for i in range(args.warmup):
        step(i)
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, events[i])
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    k_ms = sum(e[0].elapsed_time(e[1]) for e in events) / args.steps
    v_ms = sum(e[1].elapsed_time(e[2]) for e in events) / args.steps
    elapsed = sharding.max_over_ranks(elapsed, dev)  # the step takes as long as the slowest rank

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = step_bytes * world / (elapsed / args.steps) / 1e9
        target_ms, target_bytes, target_name = (v_ms, bytes_v, f"{vk} dequant (V set)")
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch from rocprofv3 PMC passes (see profiles/README.md)
            try:
                traffic = json.load(open(tpath)).get(args.workload, {}).get(f"dequant_{vk}")
            except Exception:
                traffic = None
        line = {
            "metric": f"KV dequant GB/s vs HBM roofline (quant_{mode} step: {kk.upper()} K + {vk.upper()} V -> fp16)",
            "value": round(value, 1),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",  # arithmetic: (float)q * scale in fp32, rounded once to the fp16 output
            "dtype_detail": "int8 / packed-int4 in, fp32 multiply, fp16 out",
            "data": "synthetic",
            "config": {"workload": args.workload, "shape_LBHTD": [L, B, H, T, D], "mode": mode,
                       "step": "QuantizedKVCache.to_past_key_values(): 2 launches (K set, V set)",
                       "bytes_per_step": int(step_bytes), "parallelism": f"batch-shard x{world}, no collective",
                       "timing_reduction_backend": backend, "rotating_caches": len(caches)},
            "roofline": {
                "kernel": f"dequant_tokens_fast_k<{vk}>", "what": target_name, "bound": "hbm",
                "achieved": round(target_bytes / (target_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(target_bytes / (target_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "traffic": traffic, "algorithmic_bytes_per_launch": int(target_bytes),
                "avg_launch_ms": round(target_ms, 4), "timer": "HIP events on the launch stream, per launch, in the timed region",
            },
            "roofline_k": {
                "kernel": f"dequant_tokens_fast_k<{kk}>", "bound": "hbm",
                "achieved": round(bytes_k / (k_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(bytes_k / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "algorithmic_bytes_per_launch": int(bytes_k), "avg_launch_ms": round(k_ms, 4),
            },
            "roofline_quantise": quant_info,
            "est_kv_cache_mb": round(est_mb, 3),
            "fp16_kv_cache_mb": round(2 * n_elts * 2 / 2**20, 3),
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(L, B, H, T, D, args.cpu_sample_layers)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
