#!/usr/bin/env python3
"""bench.py — the BASELINE metric of the MI355X KV-cache quantise / dequantise / eviction path in ONE line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload llama3_8b_mixed_seq16k]

Headline workload (BASELINE.json configs[3], the one the target is quoted on): Llama-3-8B KV shape
[L=32, 2, B=1, H_kv=8, T=16384, D=128], ``quant_mixed`` = INT8 keys + packed-INT4 values, synthetic N(0,1)
fp16 KV (seed 42) quantised once by the HIP quantise kernels before timing.

One STEP = dequantise the whole cache, all 32 layers of K (INT8 -> fp16) and V (INT4 -> fp16) — what the
reference does before every decode forward (reference src/quantization/ops.py:345-355,
src/benchmarking/benchmarker.py:470): exactly two kernel launches into pre-allocated rotating outputs
(``_KVStore.dequant(out=...)``, the store-level call ``QuantizedKVCache.to_past_key_values(copy=True)`` makes
after allocating its two output tensors; the public call itself is timed beside it: ``public_api``).

value    = algorithmic bytes of the step (INT8 3.0 B/elt, INT4 2.5 B/elt) x ranks / max-over-ranks
           wall time, GB/s. Weak scaling: every rank holds its own prompt's cache (batch shard, no collective).
roofline = the INT4 dequantise kernel (north-star kernel): algorithmic bytes per launch (1,342,177,280) / its mean
           duration over the timed region, from HIP events bound to each launch's own dispatch; peak 8000 GB/s.
           ``roofline.kernel`` (and every other ``kernel`` field) is what the library launched, as the profiler
           prints it (kvq_kernel_log), never a name typed here.
Sub-records of the default line (the rest of BASELINE.json's metric; each with its own small fixed iteration count):
  decode                          configs[1]: gpt2 quant_int8, prompt 512 + 512 new tokens through
                                  KVCacheBenchmarker.benchmark_method: tokens/sec + est_kv_cache_mb for the staged
                                  (bit-exact), fused-attention and HIP-graph decode beside full_cache
  configs.gpt2m_int4_seq4k        configs[2] made HBM-bound by rotating 4 caches: dequantise + quantise rooflines
                                  at the 16-row x 64 shape
  configs.gpt2_shape_seq32k       the default model's 12-row x 64 shape at an HBM-sized T (per-shape table)
  configs.llama3_8b_evict_seq32k  configs[4] per-GPU share: chunk mean-pool + sliding-window rooflines
  sharded_quant (N > 1 only)      the ONE data-path collective: batch-sharded quantise with all_reduce(MAX) over RCCL
cpu_baseline = the C restatement of the reference's algorithm (oracle/kvq_oracle.c, scalar, 1 core), the
           whole-tensor torch-CPU form and the reference's literal per-slice loop on bounded samples of the same
           workload, on this box's host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

_T_PROCESS0 = time.perf_counter()  # before `import torch`: the start-up share of `phases_s`

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: (L, B, H, T, D, mode)
    "llama3_8b_mixed_seq16k": (32, 1, 8, 16384, 128, "mixed"),
    "gpt2m_int4_seq4k": (24, 1, 16, 4096, 64, "int4"),
    "gpt2_int8_seq1k": (12, 1, 12, 1024, 64, "int8"),
}
# shapes measured as sub-records of the default line: (L, B, H, T, D, config mode, rotating caches)
SHAPE_RECORDS = {
    "gpt2m_int4_seq4k": (24, 1, 16, 4096, 64, "int4", 4),     # BASELINE configs[2]; 96 MiB store: rotation makes it HBM-bound
    "gpt2_shape_seq32k": (12, 1, 12, 32768, 64, "int8", 2),   # the default model's head shape at an HBM-sized T
}
# BASELINE.json configs[1]/[2] as decode loops through KVCacheBenchmarker (random-init weights of
# the named architecture; see benchmarking/offline.py): --workload decode:<arch>:<method>
DECODE_DEFAULT = ("gpt2", "quant_int8", 512, 512)  # arch, method, prompt tokens, new tokens
# BASELINE.json configs[4]: Llama-3-8B sliding_window + chunk_summary, seq 32K, batch 64 sharded
# over 8 GPUs = 8 batch rows per GPU: per-rank KV [L=32, 2, B=8, H=8, T=32768, D=128] fp16 = 32 GiB
EVICT = {"llama3_8b_evict_seq32k": (32, 8, 8, 32768, 128, 256, 64, 256)}  # L,B,H,T,D,window,chunk,keep_last
# scope row N3 at the same per-GPU share: the index-select policies (one row-gather launch over the 64-tensor tuple each, the
# reference's benchmark_method defaults: benchmarker.py:643-660) and PagedKVCache.get_kv of one layer; runs on EVICT's tensors
SPARSE = {"llama3_8b_sparse_seq32k": ("llama3_8b_evict_seq32k", dict(window_size=256, block_size=64, prefix_len=32, stride=4, keep_per_block=8, old_budget=64))}
# SURVEY §8e's one exchange step: ONE batched [B=64,H,T,D] slice per layer whose batch rows are split over the ranks
# (strong scaling: the global batch is fixed): abs-max of the local rows -> all_reduce(MAX) of the [G,T] table ->
# quantise with the whole batch's scales. T = 1 is a decode step's append, T = 512 a prefill chunk.
SHARDQ = {  # name: (L, B_global, H, T, D)
    "llama3_8b_batch64_sharded_append": (32, 64, 8, 1, 128),
    "llama3_8b_batch64_sharded_prefill512": (32, 64, 8, 512, 128),
}
# scope row N1 (second form): one decode step's attention over the quantised store, all layers
ATTN = {  # name: (L, B, Hq, Hkv, T, D, mode)
    "llama3_8b_decode_attn_seq16k": (32, 1, 32, 8, 16384, 128, "mixed"),
    "llama3_8b_decode_attn_seq16k_b8": (32, 8, 32, 8, 16384, 128, "mixed"),
    "gpt2_decode_attn_seq1k": (12, 1, 12, 12, 1024, 64, "int8"),
    "llama3_8b_decode_attn_seq1k": (32, 1, 32, 8, 1024, 128, "mixed"),      # short contexts: launch-bound (two kernels per layer call)
    "llama3_8b_decode_attn_seq2k_b8": (32, 8, 32, 8, 2048, 128, "mixed"),
    "llama2_7b_decode_attn_seq4k_b8": (32, 8, 32, 32, 4096, 128, "mixed"),  # multi-head (one query head per kv head)
    "llama32_1b_decode_attn_seq16k_b8": (16, 8, 32, 8, 16384, 64, "mixed"),  # grouped-query at head_dim 64
}
QUANT_ROW_PAD_TOKENS = 48  # profiles/r04a_quant_stride_table.md: any pad >= 4 tokens moves the input's head rows off a power-of-two stride
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
BYTES_PER_ELT = {"int8": 3.0, "int4": 2.5}  # q read + fp16 write
MODE_KINDS = {"int8": ("int8", "int8"), "int4": ("int4", "int4"), "mixed": ("int8", "int4")}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="llama3_8b_mixed_seq16k",
                    help="one of %s, or decode:<arch>:<method>[:<prompt_tokens>:<new_tokens>] "
                         "(e.g. decode:gpt2:quant_int8:512:512; steps = prompts per rank), or shape:<%s>" % (sorted(list(WORKLOADS) + list(ATTN) + list(EVICT) + list(SPARSE) + list(SHARDQ)), "|".join(SHAPE_RECORDS)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-subrecords", action="store_true",
                    help="headline workload only: skip the decode / configs / sharded_quant sub-records of the default line "
                         "(profiling runs that want nothing but the two dequantise kernels)")
    ap.add_argument("--rotate-caches", type=int, default=2,
                    help="independent quantised caches visited round-robin by consecutive steps, so that no "
                         "step re-reads lines the 256 MiB Infinity Cache may still hold (1 = the decode loop's "
                         "behaviour: the same cache every step; its 256 MiB INT4 store then stays cache-resident)")
    ap.add_argument("--cpu-sample-layers", type=int, default=4)
    ap.add_argument("--cpu-sample-tokens", type=int, default=0,
                    help="cpu_baseline on the first N tokens of the sampled layers (0 = the whole sequence); the samples say so")
    ap.add_argument("--cpu-reps", type=int, default=5, help="timed repetitions per cpu_baseline entry (median reported)")
    ap.add_argument("--tunable", action="append", default=[], metavar="KEY=VALUE",
                    help="kvq_set_tunable(KEY, VALUE) before the run (include/kvq_hip.h lists the keys; A-B keys need "
                         "KVQ_HIP_LIB=<pkg>/lib/ab/libkvq_hip.so)")
    ap.add_argument("--per-layer-calls", action="store_true",
                    help="decode-attention workloads: one host call per layer (kvq_decode_attn) instead of one per step "
                         "(kvq_decode_step_layers)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="allow more ranks than GPUs (rank r uses GPU r %% n_gpus): a rehearsal of the N > 1 code "
                         "path on a small box, never a scaling number; RCCL refuses duplicate devices, so this "
                         "also needs --allow-gloo-timing")
    ap.add_argument("--allow-gloo-timing", action="store_true",
                    help="if RCCL cannot be brought up on every rank, run the timing / counter reductions over gloo "
                         "instead of failing (recorded as timing_reduction_backend)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds the self-launcher waits for its ranks")
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--graph-decode", action="store_true",
                    help="decode:* workloads with quant_* methods: capture the fused-attention decode step into a HIP graph "
                         "and replay it per token (implies --fused-attention)")
    ap.add_argument("--fused-attention", action="store_true",
                    help="decode:* workloads with quant_* methods: attend straight over the quantised store "
                         "(kvq_decode_attn) instead of the staged fp16 copy")
    return ap.parse_args()


# --------------------------------------------------------------------------------------------- small helpers

def _median_time(fn, reps):
    """median wall time of `reps` calls after one warm-up call"""
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def _kernels_of(fn):
    """the kernel(s) `fn` makes the library launch, as the profiler names them (kvq_kernel_log)"""
    from efficient_llm_inference_amd import _lib
    _lib.kernel_log_clear()
    fn()
    names = _lib.kernel_log()
    return names[0] if len(names) == 1 else " + ".join(names)


def _time_launches(fn, n, warm=2):
    """`fn(i)` makes exactly one library launch: its dispatch's own start / stop timestamps (kvq_time_next_launch ->
    hipExtLaunchKernelGGL), n times after `warm` untimed calls -> sorted list of milliseconds"""
    from efficient_llm_inference_amd import _lib
    for i in range(warm):
        fn(i)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(n)]
    for e in evs:  # torch creates the underlying event at its first record()
        e[0].record()
        e[1].record()
    torch.cuda.synchronize()
    for i in range(n):
        with _lib.timed_launch(evs[i][0], evs[i][1]):
            fn(warm + i)
    torch.cuda.synchronize()
    return sorted(e[0].elapsed_time(e[1]) for e in evs)


def _roofline(kernel, nbytes, ms_sorted, timer, **extra):
    avg = sum(ms_sorted) / len(ms_sorted)
    r = {"kernel": kernel, "bound": "hbm", "achieved": round(nbytes / (avg * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": round(nbytes / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None, "algorithmic_bytes_per_launch": int(nbytes),
         "avg_launch_ms": round(avg, 4), "median_launch_ms": round(ms_sorted[len(ms_sorted) // 2], 4),
         "min_launch_ms": round(ms_sorted[0], 4), "max_launch_ms": round(ms_sorted[-1], 4), "launches_timed": len(ms_sorted),
         "timer": timer}
    r.update(extra)
    return r


_DISPATCH_TIMER = "HIP events bound to each launch's own dispatch (hipExtLaunchKernelGGL start / stop timestamps)"


def _traffic(workload, key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), or None"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(workload, {}).get(key)
    except Exception:
        return None


def _device_state(step, n_steps=600, what="headline steps"):
    """Clocks / power / temperatures of the card WHILE it runs `n_steps` of the headline step (enqueued first, read while the
    queue drains): the numbers that differ between two boxes of the pool when the same stream kernel lands 7 % apart
    (pool 0.74 on one box, 0.78 on another). `rocm-smi --json` in a child process (sysfs reads, no compute
    context) + torch's amdsmi-backed readers; every failure is reported, none is fatal."""
    import subprocess
    rec = {"what": f"read while {n_steps} {what} were queued on the GPU (outside the timed region)"}
    t0 = time.perf_counter()
    for i in range(n_steps):
        step(i)
    rec["enqueue_s"] = round(time.perf_counter() - t0, 3)
    try:
        idx = torch.cuda.current_device()
        out = subprocess.run(["rocm-smi", "-d", str(idx), "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--json"],
                             capture_output=True, text=True, timeout=20).stdout
        card = next(iter(json.loads(out).values()))
        for key, val in card.items():
            k = key.lower()
            if "clock speed" in k:
                rec[key.split()[0] + "_mhz"] = int("".join(ch for ch in val if ch.isdigit()) or 0)
            elif "temperature" in k or "power (w)" in k or "performance level" in k:
                rec[key] = val
    except Exception as exc:  # noqa: BLE001 - a missing tool must not cost the bench line
        rec["rocm_smi_error"] = repr(exc)[:160]
    for name in ("clock_rate", "power_draw", "temperature", "utilization"):
        try:
            rec["torch_" + name] = getattr(torch.cuda, name)()
        except Exception as exc:  # noqa: BLE001
            rec["torch_" + name] = repr(exc)[:80]
    still_busy = not torch.cuda.current_stream().query()
    torch.cuda.synchronize()
    rec["gpu_still_busy_when_read"] = bool(still_busy)
    rec["read_s"] = round(time.perf_counter() - t0, 3)
    return rec


def _free():
    import gc
    gc.collect()
    torch.cuda.empty_cache()


# --------------------------------------------------------------------------------------------- cpu baseline

def cpu_baseline(L, B, H, T, D, sample_layers, reps=5, gpu_check=None):
    """the reference's CPU algorithm for the INT4 dequantise (the roofline kernel's
    op) and quantise of the same V set, on this box's host cores, three ways — each one warm-up +
    `reps` (>= 5 by default, every entry) timed repetitions, median reported, on a bounded sample (stated per entry):
      port        scalar C restatement (oracle/kvq_oracle.c): on 1 core, and the same loops cut into token ranges over
                  one pthread per USABLE core (`port.all_cores`) — the fair CPU ceiling
      vectorised  whole-tensor torch-CPU ops (oracle/vectorised_torch.py), torch threads = usable cores
      literal     the reference's own call structure: one op chain per [B,H,1,D] slice + a T-way
                  cat (oracle/literal_loop.py), torch-CPU, torch threads = usable cores
    plus the two eviction ops, as the reference writes them (whole-tensor torch ops) and as the C port, on one [B,H,T,D] tensor.
    USABLE cores = min(os.cpu_count(), affinity mask, cgroup CFS quota) (oracle/hostcpu.py): a pool sized to the machine's
    256 cores under a quota of 16 spends the period's budget in a few milliseconds and is frozen until the next 100 ms
    period — round 3's 100.0 ms / 999.3 ms medians. `host_cpu` records the three limits and the cgroup's throttle counters
    around the run; `timer_quantum_suspects` lists any median that still sits within 1 % of a multiple of 100 ms.
    The top-level value / cores / kind / sample are the 1-core port's dequantise figure.
    gpu_check: host copies of layer 0 as the GPU quantised / dequantised / evicted it in this run (x_k, q_k, s_k, out_k, the
    same for v, evict_x / evict_pool / evict_window, kinds): the port recomputes every op from the same bytes and reports
    bit-exactness and the largest relative difference per op (`parity`; `max_rel_err_vs_gpu` = the INT4 dequantise's —
    the accuracy gate of SURVEY §8d, expected 0.0)."""
    import numpy as np
    import torch as _t
    from oracle import c_oracle as C
    from oracle import hostcpu
    from oracle import literal_loop as LL
    from oracle import vectorised_torch as VT
    hc = hostcpu.usable_cores()
    cores = hc["usable"]
    thr0 = hostcpu.throttle_counters()
    rng = np.random.default_rng(42)
    gbps = lambda n, dt, kind="int4": round(n * BYTES_PER_ELT[kind] / dt / 1e9, 5)  # noqa: E731
    medians = {}    # entry -> median seconds, for the timer-quantum check
    throttled = {}  # entry -> CFS periods in which this cgroup was throttled while the entry was being timed (None: unreadable)

    def med(name, fn):
        t0 = hostcpu.throttle_counters()
        medians[name] = _median_time(fn, reps)
        t1 = hostcpu.throttle_counters()
        throttled[name] = None if t0 is None or t1 is None else t1[1] - t0[1]
        return medians[name]

    torch_threads = {}  # entry -> {"used": n, "probe_ms": {threads: one timed call}}

    def tmed(name, fn):
        """a torch-CPU entry: one probe call at 1 thread and one at every usable core (after a warm-up each), then the `reps`
        timed repetitions at the faster setting — an op chain below torch's parallel grain only pays for a pool (OpenMP
        wake-ups are scheduler-tick sized on a shared host), a whole-tensor pass gains from it; the entry records which"""
        probe = {}
        for th in ([1, cores] if cores > 1 else [1]):
            _t.set_num_threads(th)
            fn()
            t0 = time.perf_counter()
            fn()
            probe[th] = time.perf_counter() - t0
        best = min(probe, key=probe.get)
        _t.set_num_threads(best)
        torch_threads[name] = {"used": best, "probe_ms": {str(k): round(v * 1e3, 3) for k, v in probe.items()}}
        return med(name, fn)

    # ---- port: scalar C, 1 core and every usable core -------------------------------------------------
    n_layers = max(1, min(sample_layers, L))
    q = rng.integers(0, 256, size=(n_layers, B, H, T, D // 2), dtype=np.uint8)
    sc = (rng.random((n_layers, T), dtype=np.float32) * 0.02 + 0.001).astype(np.float32)
    n = n_layers * B * H * T * D
    # (sample sizes are chosen so that no entry's expected time sits near the 100 ms CFS period on the boxes seen so far: 3 layers for the
    # port's quantise / INT8 entries — 140-310 ms on one core —, 1 layer for the torch whole-tensor entries — 35-75 ms)
    nq_layers = max(1, min(3, n_layers))
    xq = (rng.standard_normal((nq_layers, B, H, T, D), dtype=np.float32)).astype(np.float16)
    nq = nq_layers * B * H * T * D
    q8 = rng.integers(-127, 128, size=(nq_layers, B, H, T, D), dtype=np.int8)

    def port_at(threads, tag):
        dt = med(f"port{tag}.dequant_int4", lambda: C.dequantize_tokens(q, sc, "int4", D, "f16", threads=threads))
        dq_s = med(f"port{tag}.quant_int4", lambda: C.quantize_tokens(xq, "int4", threads=threads))
        d8 = med(f"port{tag}.dequant_int8", lambda: C.dequantize_tokens(q8, sc[:nq_layers], "int8", D, "f16", threads=threads))
        dq8 = med(f"port{tag}.quant_int8", lambda: C.quantize_tokens(xq, "int8", threads=threads))
        return {"value": gbps(n, dt), "unit": "GB/s", "cores": threads, "reps": reps, "kind": "port",
                "sample": f"INT4->fp16 dequantise of {n_layers}/{L} layers of the V set [{n_layers},{B},{H},{T},{D}] "
                          f"({n} elements, median {dt:.3f} s), oracle/kvq_oracle.c scalar loops" + (f" over {threads} pthreads (token ranges)" if threads > 1 else ""),
                "quantise_value": gbps(nq, dq_s),
                "quantise_sample": f"fp16->INT4 per-token quantise of {nq_layers}/{L} layers ({nq} elements, median {dq_s:.3f} s)",
                "int8": {"dequantise_value": gbps(nq, d8, "int8"), "quantise_value": gbps(nq, dq8, "int8"),
                         "sample": f"{nq_layers}/{L} layers of the K set ({nq} elements; medians {d8:.3f} / {dq8:.3f} s)"}}

    port = port_at(1, "1")
    port["all_cores"] = port_at(cores, "N") if cores > 1 else None
    parity = None
    if gpu_check is not None:
        # SURVEY §8d's accuracy gate, op by op: what the GPU produced in this run for layer 0 against the port on the same
        # bytes (expected: bit-exact everywhere; max_rel_err is over the dequantised / pooled values)
        from oracle import kvq_oracle as O

        def rel(got, ref):
            g32, r32 = got.astype(np.float32), ref.astype(np.float32)
            return float(np.max(np.abs(g32 - r32) / np.maximum(np.abs(r32), np.float32(1e-30))))

        parity = {}
        for name, kind in (("k", gpu_check["kinds"][0]), ("v", gpu_check["kinds"][1])):
            x, gq, gs, gout = (gpu_check[f"{key}_{name}"] for key in ("x", "q", "s", "out"))
            rq, rs = C.quantize_tokens(x, kind, threads=cores)
            rdq = C.dequantize_tokens(gq, gs, kind, D, "f16", threads=cores)
            parity[f"quantise_{kind}"] = {"bit_exact": bool(np.array_equal(rq.view(np.uint8), gq.view(np.uint8)) and np.array_equal(rs.view(np.uint32), gs.view(np.uint32))),
                                          "max_rel_err": rel(C.dequantize_tokens(rq, rs, kind, D, "f16", threads=cores), rdq)}
            parity[f"dequantise_{kind}"] = {"bit_exact": bool(np.array_equal(gout.view(np.uint16), rdq.view(np.uint16))), "max_rel_err": rel(gout, rdq)}
        xe = gpu_check["evict_x"]
        rp = O.chunk_summarize_kv(xe, 64, 256)
        parity["chunk_summarize_kv"] = {"bit_exact": bool(np.array_equal(gpu_check["evict_pool"].view(np.uint16), rp.view(np.uint16))),
                                        "max_rel_err": rel(gpu_check["evict_pool"], rp)}
        rw = O.trim_kv_sliding_window(xe, 256)
        parity["trim_kv_sliding_window"] = {"bit_exact": bool(np.array_equal(gpu_check["evict_window"].view(np.uint16), rw.view(np.uint16))),
                                            "max_rel_err": rel(gpu_check["evict_window"], rw)}
        parity["sample"] = (f"layer 0 of the K and V sets as the GPU quantised / dequantised them in this run ({gpu_check['x_k'].size} elements each), "
                            f"eviction on its first {xe.shape[-2]} tokens; reference = oracle/kvq_oracle.c / kvq_oracle.py")
        vk = gpu_check["kinds"][1]
        port["max_rel_err_vs_gpu"] = parity[f"dequantise_{vk}"]["max_rel_err"]
        port["max_rel_err_sample"] = parity["sample"]

    # ---- vectorised: whole-tensor torch-CPU, usable cores ------------------------------------------
    threads_before = _t.get_num_threads()
    nv_layers = 1
    xv = _t.from_numpy(xq[:nv_layers])
    qv, sv = VT.quantize_tokens(xv, "int4")
    dv = tmed("vectorised.dequant_int4", lambda: VT.dequantize_tokens(qv, sv, "int4", D, _t.float16))
    dvq = tmed("vectorised.quant_int4", lambda: VT.quantize_tokens(xv, "int4"))
    nvv = nv_layers * B * H * T * D
    qv8, sv8 = VT.quantize_tokens(xv, "int8")
    dv8 = tmed("vectorised.dequant_int8", lambda: VT.dequantize_tokens(qv8, sv8, "int8", D, _t.float16))
    dvq8 = tmed("vectorised.quant_int8", lambda: VT.quantize_tokens(xv, "int8"))
    vect = {"value": gbps(nvv, dv), "unit": "GB/s", "cores": cores, "reps": reps, "kind": "port",
            "sample": f"INT4->fp16 dequantise of {nv_layers}/{L} layers [{nv_layers},{B},{H},{T},{D}] as whole-tensor "
                      f"torch-CPU ops ({nvv} elements, median {dv:.3f} s), oracle/vectorised_torch.py",
            "quantise_value": gbps(nvv, dvq), "quantise_sample": f"same tensors, fp16->INT4 (median {dvq:.3f} s)",
            "int8": {"dequantise_value": gbps(nvv, dv8, "int8"), "quantise_value": gbps(nvv, dvq8, "int8")},
            "torch_threads": {k.split(".")[1]: v for k, v in torch_threads.items() if k.startswith("vectorised.")},
            "note": "a chain of ~10 whole-tensor ops, each a full pass over an fp32 / int16 intermediate: ~10x the port's memory "
                    "traffic, so it can sit below the all-core port"}

    # ---- literal: the reference's per-slice loop, sub-sampled in T -------------------------------
    Ts = min(T, 1024)
    xl = _t.from_numpy(xq[0, :, :, :Ts].copy())
    qs, ss = LL.quantize_slices(xl, "int4")
    dl = tmed("literal.dequant_int4", lambda: LL.dequantize_slices(qs, ss, "int4", D, _t.float16))
    dlq = tmed("literal.quant_int4", lambda: LL.quantize_slices(xl, "int4"))
    n_lit = B * H * Ts * D
    qs8, ss8 = LL.quantize_slices(xl, "int8")
    dl8 = tmed("literal.dequant_int8", lambda: LL.dequantize_slices(qs8, ss8, "int8", D, _t.float16))
    dlq8 = tmed("literal.quant_int8", lambda: LL.quantize_slices(xl, "int8"))
    lit = {"value": gbps(n_lit, dl), "unit": "GB/s", "cores": cores, "reps": reps, "kind": "port",
           "int8": {"dequantise_value": gbps(n_lit, dl8, "int8"), "quantise_value": gbps(n_lit, dlq8, "int8")},
           "sample": f"per-slice dequantise + {Ts}-way cat of one layer's V [{B},{H},{Ts},{D}] (T sub-sampled {Ts}/{T}; the "
                     f"loop is linear in T) with torch-CPU ops (median {dl:.3f} s), oracle/literal_loop.py",
           "quantise_value": gbps(n_lit, dlq), "quantise_sample": f"same slices, per-slice fp16->INT4 (median {dlq:.3f} s)",
           "torch_threads": {k.split(".")[1]: v for k, v in torch_threads.items() if k.startswith("literal.")},
           "note": "1,024-element slices: every op is below torch's parallel grain, so the loop is single-threaded dispatch overhead whatever the pool size"}

    # ---- eviction: the reference's own whole-tensor op chains, and the C port, on ONE [B,H,T,D] tensor ---------------
    xe = _t.from_numpy(xq[0])
    W, chunk, keep = 256, 64, 256
    tw = tmed("eviction.window_torch", lambda: VT.trim_kv_sliding_window(xe, W))
    tp = tmed("eviction.pool_torch", lambda: VT.chunk_summarize_kv(xe, chunk, keep))
    tp1 = med("eviction.pool_port1", lambda: C.chunk_summarize(xq[0], chunk, keep, threads=1))
    tpn = med("eviction.pool_portN", lambda: C.chunk_summarize(xq[0], chunk, keep, threads=cores)) if cores > 1 else None
    Tout = (max(T - keep, 0) + chunk - 1) // chunk + min(keep, T)
    pool_bytes = 2.0 * B * H * D * (T + Tout)
    evict = {"cores": cores, "reps": reps, "pool_reps": reps, "kind": "port", "unit": "GB/s",
             "window_value": round(4.0 * B * H * min(W, T) * D / tw / 1e9, 5),
             "pool_value": round(pool_bytes / tp / 1e9, 5),
             "pool_port": {"cores_1": round(pool_bytes / tp1 / 1e9, 5), f"cores_{cores}": None if tpn is None else round(pool_bytes / tpn / 1e9, 5),
                           "sample": f"oracle/kvq_oracle.c chunk summary of the same tensor (medians {tp1 * 1e3:.2f} ms on 1 core"
                                     + (f", {tpn * 1e3:.2f} ms on {cores}" if tpn is not None else "") + ")"},
             "torch_threads": {k.split(".")[1]: v for k, v in torch_threads.items() if k.startswith("eviction.")},
             "sample": f"trim_kv_sliding_window(W={W}) + materialise and chunk_summarize_kv(chunk={chunk}, keep_last={keep}) of one "
                       f"[{B},{H},{T},{D}] fp16 tensor as the reference's torch ops (medians {tw * 1e3:.3f} ms / {tp * 1e3:.2f} ms), "
                       f"oracle/vectorised_torch.py"}
    _t.set_num_threads(threads_before)
    thr1 = hostcpu.throttle_counters()
    hc["throttle_counters_before_after"] = [thr0, thr1]
    hc["nr_throttled_during_baseline"] = None if thr0 is None or thr1 is None else thr1[1] - thr0[1]
    # A median that lands on a multiple of the 100 ms CFS period WHILE the cgroup was being throttled is a scheduler quantum,
    # not a rate (round 3's 100.0 / 999.3 ms). A median that merely happens to be near one (a 99 ms quantise of 33 M elements)
    # with no throttled period during its timing is listed separately, with that evidence.
    near = {k: v for k, v in medians.items() if v >= 0.095 and abs(v * 10 - round(v * 10)) <= 0.01 * round(v * 10)}
    suspects = [f"{k}: {v * 1e3:.2f} ms, {throttled[k]} throttled periods while timed" for k, v in near.items() if throttled.get(k) is None or throttled[k] > 0]
    hc["near_100ms_multiples_without_throttling"] = [f"{k}: {v * 1e3:.2f} ms" for k, v in near.items() if throttled.get(k) == 0]
    hc["entries_timed_while_throttled"] = {k: n for k, n in throttled.items() if n}
    out = {"value": port["value"], "unit": "GB/s", "cores": 1, "kind": "port", "sample": port["sample"], "reps": reps,
           "host_cores_available": cores, "host_cpu": hc, "timer_quantum_suspects": suspects,
           "port": port, "vectorised": vect, "literal": lit, "eviction": evict}
    if parity is not None:
        out["parity"] = parity
        out["max_rel_err_vs_gpu"] = port["max_rel_err_vs_gpu"]
    return out


# --------------------------------------------------------------------------------------------- decode (configs[1])

def measure_decode(arch, method, n_prompt, n_new, prompts_per_rank, warmup, world, fused=False, graph=False, extras=True):
    """BASELINE configs[1]: decode tokens/sec + KV-cache MB through KVCacheBenchmarker.benchmark_method
    (reference benchmarker.py:643-832), prompts sharded over ranks, counters aggregated once. Returns the record."""
    import efficient_llm_inference_amd as E
    from efficient_llm_inference_amd import sharding
    from efficient_llm_inference_amd.benchmarking.offline import load_model
    model, tok = load_model(arch, "cuda", torch.float16)
    bm = E.KVCacheBenchmarker(model, tok, device="cuda")
    bm.fused_attention = bool(fused or graph)
    bm.graph_decode = bool(graph)
    prompts = [f"<{n_prompt}>"] * (prompts_per_rank * world)
    for _ in range(warmup):
        bm.benchmark_method([f"<{min(n_prompt, 64)}>"], method=method, max_new_tokens=8)
    torch.cuda.synchronize()
    sharding.barrier()
    res = sharding.benchmark_sharded(bm, prompts, method, max_new_tokens=n_new)
    base = sharding.benchmark_sharded(bm, prompts, "full_cache", max_new_tokens=n_new)
    extra = {}
    if extras and method.startswith("quant_") and not bm.fused_attention:
        # the same method with the model attending over the store directly, eager and as a replayed HIP graph
        # (tolerance-level logits instead of the bit-exact staged path: reported beside `value`, never as it)
        try:
            bm.fused_attention = True
            extra["fused_attention_tokens_per_sec"] = round(sharding.benchmark_sharded(bm, prompts, method, max_new_tokens=n_new)["tokens_per_sec"], 2)
            bm.graph_decode = True
            extra["graph_decode_tokens_per_sec"] = round(sharding.benchmark_sharded(bm, prompts, method, max_new_tokens=n_new)["tokens_per_sec"], 2)
        except RuntimeError as exc:  # head_dim / attention variant the fused kernel does not serve
            extra["fused_attention_unavailable"] = str(exc)[:200]
        finally:
            bm.fused_attention = bool(fused or graph)
            bm.graph_decode = bool(graph)
    cfg = model.config
    n_layers = getattr(cfg, "num_hidden_layers", None) or cfg.n_layer
    n_heads = getattr(cfg, "num_attention_heads", None) or cfg.n_head
    kv_heads = getattr(cfg, "num_key_value_heads", None) or n_heads
    head_dim = getattr(cfg, "head_dim", None) or cfg.hidden_size // n_heads
    rec = {
        "workload": f"decode:{arch}:{method}:{n_prompt}:{n_new}", "tokens_per_sec": round(res["tokens_per_sec"], 2),
        "est_kv_cache_mb": round(res["est_kv_cache_mb_avg"], 3),
        "full_cache_tokens_per_sec": round(base["tokens_per_sec"], 2),
        "full_cache_kv_mb": round(2 * n_layers * kv_heads * head_dim * (n_prompt + n_new) * 2 / 2**20, 3),
        "vs_full_cache": round(res["tokens_per_sec"] / base["tokens_per_sec"], 3),
        "elapsed_sec": round(res["elapsed_sec"], 4), "total_new_tokens": int(res["total_new_tokens"]),
        "gpu_peak_mb": res["gpu_peak_mb"], **extra,
        "what": "KVCacheBenchmarker.benchmark_method dict keys tokens_per_sec / est_kv_cache_mb_avg (reference benchmarker.py:811-832); "
                "tokens_per_sec = the staged decode (bit-exact with the reference's loop), fused / graph = attention over the store",
        "config": {"arch": arch, "method": method, "fused_attention": bm.fused_attention, "graph_decode": bm.graph_decode,
                   "prompt_tokens": n_prompt, "new_tokens": n_new, "prompts_per_rank": prompts_per_rank,
                   "weights": "random-init (offline)", "layers": n_layers, "heads": n_heads, "kv_heads": kv_heads, "head_dim": head_dim},
    }
    del bm, model
    _free()
    return rec


def run_decode(args, rank, world, dev):
    from efficient_llm_inference_amd import sharding
    parts = args.workload.split(":")
    arch = parts[1] if len(parts) > 1 else DECODE_DEFAULT[0]
    method = parts[2] if len(parts) > 2 else DECODE_DEFAULT[1]
    n_prompt = int(parts[3]) if len(parts) > 3 else DECODE_DEFAULT[2]
    n_new = int(parts[4]) if len(parts) > 4 else DECODE_DEFAULT[3]
    rec = measure_decode(arch, method, n_prompt, n_new, args.steps, args.warmup, world, args.fused_attention, args.graph_decode)
    if rank == 0:
        cfg = rec.pop("config")
        print(json.dumps({
            "metric": "decode tokens/sec + KV-cache MB", "value": rec["tokens_per_sec"], "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(rec["elapsed_sec"] / max(1, args.steps) * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16 model, u8/u4 KV", "data": "synthetic",
            "config": {"workload": args.workload, **cfg, "parallelism": f"prompt-shard x{world}, one all_reduce of counters",
                       "timing_reduction_backend": sharding.backend()},
            **{k: v for k, v in rec.items() if k not in ("workload", "tokens_per_sec", "what")},
        }), flush=True)


# --------------------------------------------------------------------------------------------- eviction (configs[4])

def _measure_sparse(past, shape, pol, iters=6):
    """Scope row N3 on tensors that are already resident: every index-select policy of the reference (implementations.py:143-292)
    through its public function = ONE row-gather launch over the whole tuple (+ the output allocation and the index upload),
    and PagedKVCache.get_kv (implementations.py:82-106) of one layer's cache. Algorithmic bytes: 4 per kept element (2 read +
    2 written); each launch timed by HIP events bound to its own dispatch."""
    from efficient_llm_inference_amd import cache as C
    from efficient_llm_inference_amd.cache import PagedKVCache
    L, B, H, T, D = shape
    W, P = pol["window_size"], pol["prefix_len"]
    calls = {
        "trim_kv_strided": lambda: C.trim_kv_strided(past, window_size=W, stride=pol["stride"], prefix_len=P),
        "trim_kv_block_old": lambda: C.trim_kv_block_old(past, window_size=W, block_size=pol["block_size"], keep_per_block=pol["keep_per_block"], prefix_len=P),
        "trim_kv_budget_old": lambda: C.trim_kv_budget_old(past, window_size=W, old_budget=pol["old_budget"], prefix_len=P),
        "trim_kv_prefix_window": lambda: C.trim_kv_prefix_window(past, prefix_len=P, window_size=W),
    }
    rec = {"shape_per_rank_L2BHTD": [L, 2, B, H, T, D], "policy_arguments": pol,
           "what": "each policy = its public function over the 64-tensor tuple: ONE kvq_gather_tokens launch; bytes = 4 per kept element"}
    for name, fn in calls.items():
        out = fn()
        kept = out[0][0].size(2)
        del out
        kern = _kernels_of(lambda: fn())
        ms = _time_launches(lambda i: fn(), iters, warm=1)
        rec[name] = _roofline(kern, 4.0 * 2 * L * B * H * kept * D, ms, _DISPATCH_TIMER, kept_tokens=kept, of_tokens=T)
        _free()
    pc = PagedKVCache(block_size=pol["block_size"], device="cuda", dtype=torch.float16)
    pc.extend(past[0][0], past[0][1])  # one layer's K and V: T / block_size blocks each
    kern = _kernels_of(lambda: pc.get_kv())
    # get_kv makes one launch per pool (K, then V): the events bind to the K launch; both move the same bytes
    ms = _time_launches(lambda i: pc.get_kv(), iters, warm=1)
    # a pool is ONE allocation: its stitch is one launch for up to 65,535 blocks (kvq_window_compact adds block * stride in the
    # kernel; before round 4's last change: 128 block pointers per launch, 4 launches per pool here); the events bind to the K
    # pool's launch; the whole call (2 allocations + both pools' launches) by wall events
    first = min(pc.num_blocks(), 65535)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    pc.get_kv()
    ev[0].record()
    for _ in range(iters):
        pc.get_kv()
    ev[1].record()
    torch.cuda.synchronize()
    call_ms = ev[0].elapsed_time(ev[1]) / iters
    rec["paged_get_kv"] = _roofline(kern, 4.0 * first * B * H * pol["block_size"] * D, ms, _DISPATCH_TIMER, blocks=pc.num_blocks(), block_size=pol["block_size"],
                                    blocks_in_the_timed_launch=first, launches_per_call=2 * -(-pc.num_blocks() // 65535),
                                    whole_call={"ms": round(call_ms, 4), "GBps": round(2 * 4.0 * B * H * T * D / (call_ms * 1e-3) / 1e9, 1),
                                                "what": "PagedKVCache.get_kv(): two output allocations + every stitch launch of the K and V pools, HIP events around the call"},
                                    what="PagedKVCache.get_kv of ONE layer: the K pool's stitch launch (every block)")
    del pc
    _free()
    worst = min((v["frac"], k) for k, v in rec.items() if isinstance(v, dict) and "frac" in v and v.get("algorithmic_bytes_per_launch", 0) > 1e9)
    rec["roofline"] = rec["trim_kv_strided"]  # the policy that moves the most bytes (a quarter of the old tokens)
    rec["lowest_fraction_above_1GB"] = {"op": worst[1], "frac": worst[0]}
    return rec


def measure_evict(name, dev, rank, world, steps, warmup, with_sparse=None):
    """configs[4] per-rank slice: one STEP = trim_kv_sliding_window + chunk_summarize_kv over the
    whole legacy tuple (64 tensors of [8,8,32768,128] fp16): two launches, inputs resident.
    with_sparse: policy arguments -> also the N3 record on the same tensors (rec["sparse"])."""
    import efficient_llm_inference_amd as E
    from efficient_llm_inference_amd import _lib, sharding
    from efficient_llm_inference_amd.kernels import chunk_summary_len
    L, B, H, T, D, W, chunk, keep = EVICT[name]
    torch.manual_seed(42 + rank)
    past = tuple((torch.randn(B, H, T, D, device=dev, dtype=torch.float16),
                  torch.randn(B, H, T, D, device=dev, dtype=torch.float16)) for _ in range(L))
    Tout = chunk_summary_len(T, chunk, keep)
    n_t = 2 * L
    bytes_win = 4.0 * n_t * B * H * W * D                      # 2 B read + 2 B write per kept element
    bytes_pool = 2.0 * n_t * B * H * D * (T + Tout)            # read every token once, write Tout rows
    step_bytes = bytes_win + bytes_pool
    k_win = _kernels_of(lambda: E.trim_kv_sliding_window(past, W))
    k_pool = _kernels_of(lambda: E.chunk_summarize_kv(past, chunk_size=chunk, keep_last=keep))

    def step(evs=None):
        if evs is None:
            E.trim_kv_sliding_window(past, W)
            E.chunk_summarize_kv(past, chunk_size=chunk, keep_last=keep)
            return
        with _lib.timed_launch(evs[0], evs[1]):
            E.trim_kv_sliding_window(past, W)
        with _lib.timed_launch(evs[2], evs[3]):
            E.chunk_summarize_kv(past, chunk_size=chunk, keep_last=keep)

    for _ in range(warmup):
        step()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
    for evs in events:
        for e in evs:
            e.record()
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(events[i])
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    w_ms = sorted(e[0].elapsed_time(e[1]) for e in events)
    p_ms = sorted(e[2].elapsed_time(e[3]) for e in events)
    elapsed = sharding.max_over_ranks(elapsed, dev)  # the step takes as long as the slowest rank
    rec = {
        "value": round(step_bytes * world / (elapsed / steps) / 1e9, 1), "unit": "GB/s", "steps": steps,
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "config": {"workload": name, "shape_per_rank_L2BHTD": [L, 2, B, H, T, D], "window": W,
                   "chunk_size": chunk, "keep_last": keep, "global_batch": B * world,
                   "step": "trim_kv_sliding_window(past, 256) + chunk_summarize_kv(past, 64, 256) over the 64-tensor legacy tuple "
                           "(public functions: output allocation + one launch each)",
                   "parallelism": f"batch-shard x{world} (8 rows per GPU), no collective",
                   "timing_reduction_backend": sharding.backend()},
        "roofline": _roofline(k_pool, bytes_pool, p_ms, _DISPATCH_TIMER, what="chunk_summarize_kv (mean-pool + recent tail)",
                              traffic=_traffic(name, "chunk_pool")),
        "roofline_window": _roofline(k_win, bytes_win, w_ms, _DISPATCH_TIMER, what="trim_kv_sliding_window, materialised"),
    }
    if rank == 0:  # the card's clocks / power / temperatures under THIS stream kernel (the one that differs most between boxes)
        rec["device_state"] = _device_state(lambda i: E.chunk_summarize_kv(past, chunk_size=chunk, keep_last=keep), n_steps=40,
                                            what="chunk_summarize_kv launches (5-6 ms each)")
    if with_sparse is not None:
        try:
            rec["sparse"] = _measure_sparse(past, (L, B, H, T, D), with_sparse)
        except Exception as exc:  # noqa: BLE001 — a sub-record never costs the record it rides on
            import traceback
            traceback.print_exc(file=sys.stderr)
            rec["sparse"] = {"error": f"{type(exc).__name__}: {exc}"[:400]}
    del past
    _free()
    return rec


def run_sparse(args, rank, world, dev):
    """--workload llama3_8b_sparse_seq32k: the N3 record alone (same tensors as the eviction workload)"""
    base, pol = SPARSE[args.workload]
    rec = measure_evict(base, dev, rank, world, 2, 1, with_sparse=pol)["sparse"]
    if rank == 0:
        r = rec["roofline"]
        print(json.dumps({
            "metric": "index-select eviction policies + paged stitch, GB/s vs HBM roofline (row gather, 4 B per kept element)",
            "value": r["achieved"], "unit": "GB/s", "n_gpus": world, "steps": r["launches_timed"], "warmup": 1,
            "ms_per_step": r["avg_launch_ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16",
            "dtype_detail": "fp16 rows moved as bytes", "data": "synthetic",
            "config": {"workload": args.workload, "shape_per_rank_L2BHTD": rec["shape_per_rank_L2BHTD"], "policy_arguments": rec["policy_arguments"],
                       "step": "trim_kv_strided over the 64-tensor tuple (one launch); the other policies and PagedKVCache.get_kv beside it",
                       "parallelism": f"batch-shard x{world}, no collective"},
            "roofline": r, **{k: v for k, v in rec.items() if k not in ("roofline", "shape_per_rank_L2BHTD", "policy_arguments")},
        }), flush=True)


def run_evict(args, rank, world, dev):
    rec = measure_evict(args.workload, dev, rank, world, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps({
            "metric": "KV eviction GB/s vs HBM roofline (sliding_window + chunk_summary step)",
            "value": rec["value"], "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "dtype_detail": "fp16 in/out, fp32 accumulate", "data": "synthetic", "config": rec["config"],
            "roofline": rec["roofline"], "roofline_window": rec["roofline_window"], "device_state": rec.get("device_state"),
        }), flush=True)


# --------------------------------------------------------------------------------------------- sharded quantise (§8e)

def measure_sharded_quant(name, dev, rank, world, steps, warmup):
    """One STEP = quantise_tokens_batch_sharded of the K set (INT8) and the V set (INT4) of every layer: per layer chunk an
    abs-max launch, an all_reduce(MAX) of its [Lc,T] fp32 table on a side stream (the ONLY collective that carries path
    data) and a quantise launch that re-reads the chunk while the Infinity Cache still holds it. Each rank holds
    B_global / world batch rows."""
    from efficient_llm_inference_amd import sharding
    L, Bg, H, T, D = SHARDQ[name]
    rows = sharding.shard_batch_rows(Bg)
    Bl = len(rows)
    torch.manual_seed(42 + rank)
    k = torch.randn(L, Bl, H, T, D, device=dev, dtype=torch.float16)
    v = torch.randn(L, Bl, H, T, D, device=dev, dtype=torch.float16)
    step_bytes = L * Bg * H * T * D * (BYTES_PER_ELT["int8"] + BYTES_PER_ELT["int4"])  # whole job, single-pass bytes
    outs = [sharding.ShardedQuantBuffers(k, "int8"), sharding.ShardedQuantBuffers(v, "int4")]

    def run(two_phase):
        def step():  # K and V together: ONE abs-max launch and ONE collective when the [2L,T] table is small (a decode append)
            sharding.quantize_kv_batch_sharded(k, v, ("int8", "int4"), outs=outs, two_phase=two_phase)

        kernels = _kernels_of(step)
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        sharding.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        sharding.barrier()
        torch.cuda.synchronize()
        return sharding.max_over_ranks(time.perf_counter() - t0, dev), kernels

    # what the call does at THIS world size: the phases + all_reduce at N > 1; at N = 1 there is nothing to exchange and
    # the slice takes the un-sharded single-pass call — the phases a rank of a larger job runs are timed beside it
    elapsed, kernels = run(None)
    phases = None
    if world == 1:
        el2, k2 = run(True)
        phases = {"value": round(step_bytes / (el2 / steps) / 1e9, 2), "unit": "GB/s", "ms_per_step": round(el2 / steps * 1e3, 4),
                  "frac_of_hbm_peak_per_gpu": round(step_bytes / (el2 / steps) / 1e9 / HBM_PEAK_GBPS, 4), "kernels": k2,
                  "what": "two_phase=True on one rank: abs-max pass, (no exchange), quantise pass — the input is read twice, "
                          "as on every rank of an N > 1 job; bytes counted once"}
    rec = {
        "value": round(step_bytes / (elapsed / steps) / 1e9, 2), "unit": "GB/s", "steps": steps,
        "ms_per_step": round(elapsed / steps * 1e3, 4), "scaling": "strong",
        "frac_of_hbm_peak_per_gpu": round(step_bytes / world / (elapsed / steps) / 1e9 / HBM_PEAK_GBPS, 4),
        "kernels": kernels,
        "config": {"workload": name, "shape_L_Bglobal_H_T_D": [L, Bg, H, T, D], "batch_rows_per_rank": Bl,
                   "step": (("K + V together: ONE kvq_absmax_tokens over the 2L tensors -> ONE all_reduce(MAX) [2L,T] fp32 -> "
                             "kvq_quant_tokens_from_absmax of K (INT8) and of V (INT4)") if sharding.kv_joint_table_ok(L, T) else
                            ("per layer chunk: kvq_absmax_tokens -> all_reduce(MAX) [Lc,T] fp32 (side stream) -> "
                             "kvq_quant_tokens_from_absmax; K (INT8) then V (INT4)")) if world > 1 else
                           "one rank, nothing to exchange: kvq_quant_i8_tokens / kvq_quant_i4_tokens, ONE pass (see two_phase)",
                   "layer_chunks": outs[0].n_chunks, "kv_joint_table": sharding.kv_joint_table_ok(L, T),
                   "collectives_per_step": 1 if sharding.kv_joint_table_ok(L, T) else 2 * outs[0].n_chunks,
                   "collective": "all_reduce(MAX)", "collective_backend": sharding.backend() or "none (1 rank)",
                   "collective_bytes_per_step": 2 * L * T * 4, "bytes_per_step_single_pass": int(step_bytes),
                   "parallelism": f"batch rows sharded x{world}; " + ("one all_reduce(MAX) per step (K + V tables joined)" if sharding.kv_joint_table_ok(L, T)
                                                                    else "one all_reduce(MAX) per layer chunk, set and step"),
                   "timing_reduction_backend": sharding.backend()},
    }
    if phases is not None:
        rec["two_phase"] = phases
    del k, v, outs
    _free()
    return rec


def run_sharded_quant(args, rank, world, dev):
    rec = measure_sharded_quant(args.workload, dev, rank, world, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps({
            "metric": "batch-sharded KV quantise step (INT8 K + INT4 V, one scale per token across the WHOLE batch), GB/s",
            "value": rec["value"], "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "config": rec["config"], "kernels": rec["kernels"],
            "frac_of_hbm_peak_per_gpu": rec["frac_of_hbm_peak_per_gpu"], "two_phase": rec.get("two_phase"),
        }), flush=True)


# --------------------------------------------------------------------------------------------- per-shape rooflines

def measure_shape(name, dev, rank, iters=24):
    """Dequantise + quantise rooflines of one KV shape, both kinds, with `n_rot` independent input sets / stores /
    outputs visited round-robin so that nothing is served from the 256 MiB Infinity Cache (a 96 MiB INT4 store
    dequantised again and again would be)."""
    from efficient_llm_inference_amd import kernels as K
    from efficient_llm_inference_amd.quantization.ops import _KVStore
    L, B, H, T, D, mode, n_rot = SHAPE_RECORDS[name]
    torch.manual_seed(42 + rank)
    n_elts = L * B * H * T * D
    xs = [torch.randn(L, B, H, T, D, device=dev, dtype=torch.float16) for _ in range(n_rot)]
    outs = [torch.empty(L, B, H, T, D, device=dev, dtype=torch.float16) for _ in range(n_rot)]
    for o in outs:
        o.zero_()
    rec = {"shape_LBHTD": [L, B, H, T, D], "config_mode": mode, "rotating_sets": n_rot, "rows_x_head_dim": f"{B * H} x {D}",
           "working_set_mb": round(n_rot * n_elts * (2 + 2 + 1.5) / 2**20, 1)}
    for kind in ("int8", "int4"):
        # the quantised stores are small (50 MB per INT4 set at config 3): rotate enough of them that the kernel's READ
        # side alone exceeds the 256 MiB Infinity Cache more than twice over (the fp16 side rotates n_rot buffers)
        store_bytes = n_elts * (1.0 if kind == "int8" else 0.5)
        n_st = max(n_rot, -(-(640 << 20) // int(store_bytes)))
        stores = []
        for i in range(n_st):
            st = _KVStore(kind, L, dev)
            st.reserve(T)
            st.append(xs[i % n_rot])
            stores.append(st)
        ws = stores[0]._workspace(L * T)
        k_q = _kernels_of(lambda: K.quant_tokens(xs[0], stores[0].q[:, :, :, :T], stores[0].scales[:, :T], ws, kind))
        k_d = _kernels_of(lambda: stores[0].dequant(torch.float16, out=outs[0]))
        q_ms = _time_launches(lambda i: K.quant_tokens(xs[i % n_rot], stores[i % n_st].q[:, :, :, :T], stores[i % n_st].scales[:, :T], ws, kind), iters)
        d_ms = _time_launches(lambda i: stores[i % n_st].dequant(torch.float16, out=outs[i % n_rot]), iters)
        nbytes = n_elts * BYTES_PER_ELT[kind]
        rec[f"dequant_{kind}"] = _roofline(k_d, nbytes, d_ms, _DISPATCH_TIMER, rotating_stores=n_st)
        rec[f"quant_{kind}"] = _roofline(k_q, nbytes, q_ms, _DISPATCH_TIMER, rotating_stores=n_st)
        # the same launch with the input's head rows off their stride ([:, :, :, :T] views of T + 48 token rows) — only where the
        # contiguous stride is a multiple of 1 MiB: 4 MiB rows are the one stride measured to hurt, at 512 KiB a pad of this size
        # costs 10 % (profiles/r04a_quant_stride_table.md)
        if (T * D * 2) % (1 << 20) == 0:
            xps = []
            for x in xs:
                full = torch.empty(L, B, H, T + QUANT_ROW_PAD_TOKENS, D, device=dev, dtype=torch.float16)
                full[:, :, :, :T].copy_(x)
                xps.append(full[:, :, :, :T])
            p_ms = _time_launches(lambda i: K.quant_tokens(xps[i % n_rot], stores[i % n_st].q[:, :, :, :T], stores[i % n_st].scales[:, :T], ws, kind), iters)
            rec[f"quant_{kind}"]["padded_rows"] = _roofline(k_q, nbytes, p_ms, _DISPATCH_TIMER, input=f"[L,B,H,T+{QUANT_ROW_PAD_TOKENS},D][:, :, :, :T] views, same values")
            del xps
        del stores
    vk = MODE_KINDS[mode][1]
    rec["roofline"] = rec[f"dequant_{vk}"]            # the config's own kind (V set)
    rec["roofline_quantise"] = rec[f"quant_{vk}"]
    del xs, outs
    _free()
    return rec


# --------------------------------------------------------------------------------------------- decode attention (N1)

def run_attn(args, rank, world, dev):
    """One STEP = the attention of one decode step over the quantised store of every layer
    (kvq_decode_attn per layer: split-T partial kernel + merge), new token's exact K/V included.
    Side measurement: the same step as the staged path runs it (torch SDPA over the fp16 copy)."""
    import efficient_llm_inference_amd as E
    from efficient_llm_inference_amd import kernels as K
    from efficient_llm_inference_amd import sharding
    L, B, Hq, Hkv, T, D, mode = ATTN[args.workload]
    kk, vk = MODE_KINDS[mode]
    torch.manual_seed(42 + rank)
    qc = E.QuantizedKVCache(n_layers=L, mode=mode, device="cuda", compute_dtype=torch.float16)
    qc.reserve(T + int(os.environ.get("KVQ_BENCH_TCAP_PAD", "0")))  # (experiment knob: store rows off their power-of-two stride)
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

    plan = K.DecodeLayersPlan(q, kn, vn, out, qc._k.q, qc._k.scales, kk, qc._v.q, qc._v.scales, vk)

    def step():
        if args.per_layer_calls:  # one trip through the binding per layer (what an HF attention hook does)
            for i in range(L):
                K.decode_attn(q[i], qc._k.q[i], qc._k.scales[i], kk, qc._v.q[i], qc._v.scales[i], vk, T, out[i], ws, sm,
                              kn[i], vn[i])
        else:  # kvq_decode_step_layers: ONE host call enqueues every layer's launch
            K.decode_step_layers(plan, T, ws, sm)

    kernels = _kernels_of(step)
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
                       "step": "one decode step = every layer's attention over the store: "
                               + ("kvq_decode_attn per layer" if args.per_layer_calls else "ONE kvq_decode_step_layers call")
                               + ", host launch gaps included",
                       "launches_per_layer": len(kernels.split(" + ")),
                       "bytes_per_step": int(step_bytes), "parallelism": f"batch-shard x{world}, no collective",
                       "timing_reduction_backend": sharding.backend()},
            "roofline": {"kernel": kernels + " (per layer)", "bound": "hbm",
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


# --------------------------------------------------------------------------------------------- launcher

def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args, argv):
    """``python bench.py --gpus N`` without a launcher: start N ranks of this script (one per GPU)
    BEFORE anything in this process touches the GPU, relay rank 0's JSON line, exit non-zero if any
    rank does. The parent never initialises HIP (``torch.cuda.device_count()`` does not on this
    image) and never re-execs itself; the children are ordinary child processes."""
    import subprocess
    n = args.gpus
    if not args.launcher_selftest:
        n_dev = torch.cuda.device_count()
        if n_dev == 0:
            raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU path")
        if n > n_dev and not args.share_gpu:
            raise SystemExit(f"bench.py: --gpus {n} but only {n_dev} GPU(s) are visible; a rehearsal with several ranks "
                             f"per GPU needs --share-gpu (and --allow-gloo-timing: RCCL refuses duplicate devices)")
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv,
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    deadline = time.monotonic() + args.launch_timeout
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc not in (None, 0):
                failed = (r, rc)
        if time.monotonic() > deadline:
            failed = (-1, 124)
        time.sleep(0.05)
    if failed is None:
        failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:  # the other ranks would wait in a collective for ever: stop exactly the processes we started
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        out0 = procs[0].stdout.read() if procs[0].stdout else ""
        sys.stderr.write(out0)
        who = "the launcher's timeout" if failed[0] < 0 else f"rank {failed[0]} (exit code {failed[1]})"
        raise SystemExit(f"bench.py --gpus {n}: failed in {who}; no result line")
    for line in procs[0].stdout.read().splitlines():  # the result line to stdout; library chatter (gloo prints to stdout) to stderr
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()


def _abort_rank(rank, world, exc):
    """A rank whose workload raised must not walk into sharding.shutdown()'s barrier (its peers sit in a collective
    of the workload: the barrier would block until the 300 s gloo timeout and bury the exception). Print the error
    and leave at once with a non-zero code; the launcher sees it and terminates the peers."""
    import traceback
    sys.stderr.write(f"bench.py: rank {rank}/{world} failed in its workload:\n")
    traceback.print_exception(type(exc), exc, exc.__traceback__, file=sys.stderr)
    sys.stderr.flush()
    sys.stdout.flush()
    os._exit(1)


def selftest_rank(args, rank, world):
    """--launcher-selftest: what a rank does with no GPU at all (CPU test of the launcher): join the
    gloo group, reduce the ranks, report."""
    from efficient_llm_inference_amd import sharding
    if os.environ.get("KVQ_SELFTEST_FAIL_RANK") == str(rank):  # test hook: a rank that dies before the rendezvous
        raise SystemExit(3)
    be = sharding.init_distributed(rank, world, None) if world > 1 else None
    if os.environ.get("KVQ_SELFTEST_RAISE_RANK") == str(rank):  # test hook: a rank whose WORKLOAD raises after the rendezvous
        _abort_rank(rank, world, RuntimeError("selftest: workload failure"))
    seen = sharding.max_over_ranks(float(rank))
    t = torch.tensor([float(rank)])
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "max_rank": seen, "rank_sum": float(t.item()),
                          "timing_reduction_backend": be}), flush=True)
    if world > 1:
        sharding.shutdown()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch as many ranks as --gpus says")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.launcher_selftest:
        return selftest_rank(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    n_dev = torch.cuda.device_count()
    share = local_rank >= n_dev or int(os.environ.get("LOCAL_WORLD_SIZE", world)) > n_dev
    if share and not args.share_gpu:
        raise SystemExit(f"bench.py: rank {rank} has no GPU of its own ({n_dev} visible, {world} ranks); "
                         f"pass --share-gpu for a rehearsal on a small box")
    dev_index = local_rank % max(1, n_dev)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import efficient_llm_inference_amd as E  # noqa: F401
    from efficient_llm_inference_amd import _lib, sharding

    backend = None
    if world > 1:
        # RCCL over xGMI carries the reductions of a run (the data path itself needs no collective).
        # RCCL not coming up on every rank is FATAL unless --allow-gloo-timing: a line that says
        # n_gpus: 8 must not hide a broken xGMI / IPC setup.
        try:
            backend = sharding.init_distributed(rank, world, dev, allow_gloo=args.allow_gloo_timing, ranks_share_device=share)
        except sharding.RcclUnavailable as exc:
            raise SystemExit(f"bench.py: {exc}")
    _lib.load()
    for kv in args.tunable:
        key, _, val = kv.partition("=")
        _lib.set_tunable(key, int(val))
    try:
        if args.workload.startswith("decode"):
            run_decode(args, rank, world, dev)
        elif args.workload in ATTN:
            run_attn(args, rank, world, dev)
        elif args.workload in EVICT:
            run_evict(args, rank, world, dev)
        elif args.workload in SPARSE:
            run_sparse(args, rank, world, dev)
        elif args.workload in SHARDQ:
            run_sharded_quant(args, rank, world, dev)
        elif args.workload.startswith("shape:") and args.workload[6:] in SHAPE_RECORDS:  # one sub-record on its own (profiling)
            rec = measure_shape(args.workload[6:], dev, rank, iters=max(4, args.steps))
            if rank == 0:
                print(json.dumps({"metric": "per-shape dequantise / quantise rooflines", "value": rec["roofline"]["achieved"], "unit": "GB/s",
                                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
                                  "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": args.workload}, **rec}), flush=True)
        elif args.workload in WORKLOADS:
            run_dequant(args, rank, world, dev, backend)
        else:
            raise SystemExit(f"unknown workload {args.workload}")
    except Exception as exc:  # noqa: BLE001
        if world > 1:
            _abort_rank(rank, world, exc)
        raise
    if world > 1:
        sharding.shutdown()


READ_CEIL_GBPS, WRITE_CEIL_GBPS = 7000.0, 6900.0  # profiles/r02ae_microbench_read_write_ceilings.txt


def _mix_bound(n_elts, kind, measured_ms):
    """Lower bound of a dequantise launch when reads and writes each run at this chip's measured ceiling and add."""
    rd = n_elts * (1.0 if kind == "int8" else 0.5)
    wr = n_elts * 2.0
    ms = (rd / READ_CEIL_GBPS + wr / WRITE_CEIL_GBPS) / 1e6
    return {"read_ceiling": READ_CEIL_GBPS, "write_ceiling": WRITE_CEIL_GBPS, "unit": "GB/s", "bound_ms": round(ms, 4),
            "frac_of_bound": round(ms / measured_ms, 4),
            "source": "profiles/r02ae_microbench_read_write_ceilings.txt (one-wave workgroups; reads and writes add)"}


def _subrecord(fn, *a, **kw):
    """a sub-record must never cost the headline: its failure is recorded in its place. At N > 1 the caller runs it under
    sharding.local_mode() — no collective inside, so a rank that fails here fails alone and nobody waits for it."""
    try:
        return fn(*a, **kw)
    except Exception as exc:  # noqa: BLE001
        import traceback
        traceback.print_exc(file=sys.stderr)
        _free()
        return {"error": f"{type(exc).__name__}: {exc}"[:400]}


def _quant_variants(tensors, kind, n_elts, dev):
    """Two more timings of the quantise launch on the same set (side measurements; scratch stores, rotating):
      padded_rows       the same values as a [:, :, :, :T] view of a [G,B,H,T+48,D] allocation — what a caller with a
                        preallocated fp16 KV buffer hands over; the 8 head rows of a tile are then 4 MiB + 12 KiB apart
                        instead of exactly 4 MiB (profiles/r04a_quant_stride_table.md: the input stride matters, the store's does not)
      outlier_channels  SURVEY §8d's second distribution: 1 % of the (head, d) channels x 8 in every token, contiguous rows —
                        times the data-dependent IEEE-divide redo of the quotient guard (kvq_quant.hip quotient_bits8)"""
    from efficient_llm_inference_amd import kernels as _k
    G = len(tensors)
    B, H, T, D = tensors[0].shape
    Dq = D if kind == "int8" else D // 2
    qdt = torch.int8 if kind == "int8" else torch.uint8
    scratch = [(torch.empty(G, B, H, T, Dq, device=dev, dtype=qdt), torch.zeros(G, T, device=dev, dtype=torch.float32)) for _ in range(2)]
    ws = torch.empty(G * T, device=dev, dtype=torch.float32)
    nbytes = n_elts * BYTES_PER_ELT[kind]
    out = {}
    full = torch.empty(G, B, H, T + QUANT_ROW_PAD_TOKENS, D, device=dev, dtype=tensors[0].dtype)
    xp = full[:, :, :, :T]
    for g, t in enumerate(tensors):
        xp[g].copy_(t)
    fn = lambda i: _k.quant_tokens(xp, scratch[i % 2][0], scratch[i % 2][1], ws, kind)  # noqa: E731
    out["padded_rows"] = _roofline(_kernels_of(lambda: fn(0)), nbytes, _time_launches(fn, 12), _DISPATCH_TIMER,
                                   input=f"[G,B,H,T+{QUANT_ROW_PAD_TOKENS},D][:, :, :, :T] view of one allocation, same values")
    del xp, full
    xo = torch.stack(list(tensors))  # contiguous [G,B,H,T,D]
    gen = torch.Generator(device="cpu").manual_seed(42)
    mask = (torch.rand(H, D, generator=gen) < 0.01).to(dev)
    xo.mul_(torch.where(mask, 8.0, 1.0).to(xo.dtype)[None, None, :, None, :])
    fn2 = lambda i: _k.quant_tokens(xo, scratch[i % 2][0], scratch[i % 2][1], ws, kind)  # noqa: E731
    out["outlier_channels"] = _roofline(_kernels_of(lambda: fn2(0)), nbytes, _time_launches(fn2, 12), _DISPATCH_TIMER,
                                        input=f"contiguous rows, {int(mask.sum())} of {H * D} (head, d) channels x 8")
    del xo, scratch, ws
    _free()
    return out


def run_dequant(args, rank, world, dev, backend):
    """The headline workload (module docstring): one STEP = dequantise the whole quantised cache (2 launches)."""
    import efficient_llm_inference_amd as E
    from efficient_llm_inference_amd import _lib as _l
    from efficient_llm_inference_amd import kernels as _k
    from efficient_llm_inference_amd import sharding
    L, B, H, T, D, mode = WORKLOADS[args.workload]
    kk, vk = MODE_KINDS[mode]
    t_run0 = time.perf_counter()
    phases = {}  # seconds of wall clock per phase of this run (rank 0's), reported as `phases_s`

    def mark(name, _t=[t_run0]):
        now = time.perf_counter()
        phases[name] = round(phases.get(name, 0.0) + now - _t[0], 2)
        _t[0] = now

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

    # two rotating output sets so consecutive steps never write the same lines (allocated before the side
    # measurement below, so that nothing idles the GPU between it, the warm-up and the timed region)
    outs = [(torch.empty(L, B, H, T, D, device=dev, dtype=torch.float16),
             torch.empty(L, B, H, T, D, device=dev, dtype=torch.float16)) for _ in range(2)]
    for ko, vo in outs:  # first touch of the fresh pages happens here, not in a timed step
        ko.zero_()
        vo.zero_()
    torch.cuda.synchronize()
    mark("build_caches")
    n_elts = L * B * H * T * D  # per K or V set
    bytes_k = n_elts * BYTES_PER_ELT[kk]
    bytes_v = n_elts * BYTES_PER_ELT[vk]
    step_bytes = bytes_k + bytes_v

    # side measurement, outside the timed region: the prefill quantise kernels (rows a1/a2) on the same tensors,
    # re-quantising into the rotating caches' stores (identical bytes every time; a 268 MB INT4 store written again
    # and again would otherwise sit in the 256 MB Infinity Cache)
    quant_info = {}
    for name, tensors in (("k", [k for k, _ in past]), ("v", [v for _, v in past])):
        stores = [getattr(c, "_" + name) for c in caches]
        ws = stores[0]._workspace(L * T)

        def qfn(i, tensors=tensors, stores=stores, ws=ws):
            st_ = stores[i % len(stores)]
            _k.quant_tokens(tensors, st_.q[:, :, :, :T], st_.scales[:, :T], ws, st_.kind)

        kern = _kernels_of(lambda: qfn(0))
        ms = _time_launches(qfn, 12)
        quant_info[f"quant_{stores[0].kind}"] = _roofline(kern, n_elts * BYTES_PER_ELT[stores[0].kind], ms, _DISPATCH_TIMER, set=name.upper(),
                                                          input="the legacy tuple: 32 contiguous [B,H,T,D] tensors, N(0,1); head rows T*D*2 B apart (4 MiB at T = 16384)")

    k_kernel = _kernels_of(lambda: caches[0]._k.dequant(torch.float16, out=outs[0][0]))
    v_kernel = _kernels_of(lambda: caches[0]._v.dequant(torch.float16, out=outs[0][1]))

    def step(i, evs=None):
        ko, vo = outs[i & 1]
        c = caches[i % len(caches)]
        c._k.dequant(torch.float16, out=ko)  # all layers of K: one launch
        if evs is not None:
            with _l.timed_launch(evs[0], evs[1]):  # the INT4 launch's own start / stop timestamps
                c._v.dequant(torch.float16, out=vo)
        else:
            c._v.dequant(torch.float16, out=vo)  # all layers of V: one launch (the roofline kernel)

    mark("quantise_side_measurement")
    for i in range(args.warmup):
        step(i)
    # Two HIP events per step, bound to the INT4 launch itself (kvq_time_next_launch -> hipExtLaunchKernelGGL): they
    # carry the dispatch's start / stop timestamps — what rocprofv3's kernel trace reports — and put no barrier
    # packets on the queue (hipEventRecord around the launch costs several microseconds of the very step whose wall
    # time is `value`, and includes the queue gap in the interval).
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.steps)]
    for e in events:  # torch creates the underlying event at its first record()
        e[0].record()
        e[1].record()
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

    mark("warmup_and_timed_steps")
    v_each = sorted(e[0].elapsed_time(e[1]) for e in events)
    elapsed = sharding.max_over_ranks(elapsed, dev)  # the step takes as long as the slowest rank

    # side measurements, outside the timed region, same rotation of caches / outputs: a longer sample of both launches
    # (the timed region's K samples are enough for a mean, thin for a min / max) ...
    v_long = _time_launches(lambda i: caches[i % len(caches)]._v.dequant(torch.float16, out=outs[i & 1][1]), 100, warm=0)
    k_long = _time_launches(lambda i: caches[i % len(caches)]._k.dequant(torch.float16, out=outs[i & 1][0]), 100, warm=0)
    # ... and the PUBLIC call the step stands for: to_past_key_values(copy=True) = two torch.empty + the same two launches
    # + the tuple of per-layer views, wall clock over 20 calls
    for i in range(2):
        caches[i % len(caches)].to_past_key_values(copy=True)
    torch.cuda.synchronize()
    tp0 = time.perf_counter()
    n_pub = 20
    for i in range(n_pub):
        pkv = caches[i % len(caches)].to_past_key_values(copy=True)
    torch.cuda.synchronize()
    pub_ms = (time.perf_counter() - tp0) / n_pub * 1e3
    del pkv
    # ... and the quantise launch on two more inputs (padded head rows, outlier channels): scratch stores, freed before the sub-records
    for name, tensors in (("k", [k for k, _ in past]), ("v", [v for _, v in past])):
        kind = getattr(caches[0], "_" + name).kind
        quant_info[f"quant_{kind}"].update(_quant_variants(tensors, kind, n_elts, dev))
    device_state = _device_state(step) if rank == 0 else None
    mark("extended_samples_and_public_api")

    gpu_check = None
    if rank == 0 and not args.no_cpu_baseline and world == 1:  # layer 0 of what the GPU just produced, for the port to check
        c0 = caches[0]
        c0._k.dequant(torch.float16, out=outs[0][0])
        c0._v.dequant(torch.float16, out=outs[0][1])
        Te = min(T, 4096)
        xe = past[0][0][:, :, :Te].contiguous()
        (pool_k, _), = E.chunk_summarize_kv(((xe, xe),), chunk_size=64, keep_last=256)
        (win_k, _), = E.trim_kv_sliding_window(((xe, xe),), 256)
        torch.cuda.synchronize()
        host = lambda t: t.contiguous().cpu().numpy()  # noqa: E731
        gpu_check = {"kinds": (kk, vk), "evict_x": host(xe), "evict_pool": host(pool_k), "evict_window": host(win_k)}
        for name, st_, o, xsrc in (("k", c0._k, outs[0][0], past[0][0]), ("v", c0._v, outs[0][1], past[0][1])):
            gpu_check[f"x_{name}"] = host(xsrc)[None]
            gpu_check[f"q_{name}"] = host(st_.q[:1, :, :, :T])
            gpu_check[f"s_{name}"] = host(st_.scales[:1, :T])
            gpu_check[f"out_{name}"] = host(o[:1])

    mark("gpu_outputs_to_host_for_parity")
    line = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = step_bytes * world / (elapsed / args.steps) / 1e9
        traffic = _traffic(args.workload, f"dequant_{vk}")
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
                       "step": "_KVStore.dequant(out=rotating buffer) of the K set and of the V set: 2 launches — the store-level calls "
                               "of QuantizedKVCache.to_past_key_values(copy=True), whose own cost is `public_api`",
                       "bytes_per_step": int(step_bytes), "parallelism": f"batch-shard x{world}, no collective",
                       "timing_reduction_backend": backend, "rotating_caches": len(caches)},
            "roofline": _roofline(
                v_kernel, bytes_v, v_each, "HIP events bound to every INT4 launch of the timed region (hipExtLaunchKernelGGL start / stop timestamps)",
                what=f"{vk} dequant (V set)", traffic=traffic,
                traffic_source="profiles/traffic.json (rocprofv3 PMC passes of this command, committed; not re-measured in this run)" if traffic is not None else None,
                # informational, beside the contract's frac: what this chip moves for THIS read / write mix. Reads and
                # writes add on its memory system; the two rates are the one-wave-workgroup ceilings measured with
                # kvq_microbench segread / fillchunk (profiles/r02ae_microbench_read_write_ceilings.txt), not this run's.
                mix_bound=_mix_bound(n_elts, vk, sum(v_each) / len(v_each)),
                extended=_roofline(v_kernel, bytes_v, v_long, _DISPATCH_TIMER + ", 100 launches after the timed region")),
            "roofline_k": _roofline(k_kernel, bytes_k, k_long, _DISPATCH_TIMER + ", 100 launches after the timed region", what=f"{kk} dequant (K set)"),
            "roofline_quantise": quant_info,
            "public_api": {"call": "QuantizedKVCache.to_past_key_values(copy=True)", "ms_per_call": round(pub_ms, 4),
                           "value": round(step_bytes / (pub_ms * 1e-3) / 1e9, 1), "unit": "GB/s", "calls_timed": n_pub,
                           "vs_raw_store_step": round(ms_per_step / pub_ms, 4),
                           "what": "wall clock of the public method (allocates 2 x 1 GiB, 2 launches, tuple of 32 (k, v) views), rotating caches"},
            "est_kv_cache_mb": round(est_mb, 3),
            "fp16_kv_cache_mb": round(2 * n_elts * 2 / 2**20, 3),
        }
    del past, caches, qc, outs
    _free()

    # ---- the rest of BASELINE.json's metric: sub-records (each guarded: a failure is recorded, never fatal) -----------
    if not args.no_subrecords and args.workload == "llama3_8b_mixed_seq16k":
        sub_t0 = time.perf_counter()
        # decode / configs: every rank measures its own share with NO collective inside (sharding.local_mode: a failure
        # is recorded and cannot hang the peers); at N > 1 the figures are rank 0's, taken while the other ranks run the same
        with sharding.local_mode():
            decode = _subrecord(measure_decode, *DECODE_DEFAULT, 1, 1, 1)
            cfgs = {}
            for name in SHAPE_RECORDS:
                cfgs[name] = _subrecord(measure_shape, name, dev, rank)
            cfgs["llama3_8b_evict_seq32k"] = _subrecord(measure_evict, "llama3_8b_evict_seq32k", dev, rank, 1, 6, 2,
                                                        with_sparse=SPARSE["llama3_8b_sparse_seq32k"][1])
            if isinstance(cfgs["llama3_8b_evict_seq32k"], dict) and "sparse" in cfgs["llama3_8b_evict_seq32k"]:
                cfgs["llama3_8b_sparse_seq32k"] = cfgs["llama3_8b_evict_seq32k"].pop("sparse")  # scope row N3, same tensors
        shq = None
        if world == 1:
            # one rank: the batch-64 prefill slice of the sharded workload, single pass beside the two phases (no collective)
            shq = _subrecord(measure_sharded_quant, "llama3_8b_batch64_sharded_prefill512", dev, rank, world, 10, 3)
        if world > 1:
            # the one data-path collective, across the ranks for real: after a rendezvous every rank reaches whatever
            # happened above; a failure in here takes the job down (rank abort), as a broken RCCL should
            torch.cuda.synchronize()
            sharding.barrier()
            shq = measure_sharded_quant("llama3_8b_batch64_sharded_prefill512", dev, rank, world, 10, 3)
        if rank == 0:
            if world > 1:
                for rec in [decode] + list(cfgs.values()):
                    if isinstance(rec, dict):
                        rec["scope"] = f"rank 0 of {world}, measured while the other ranks run the same workload (no collective inside)"
            line["decode"] = decode
            line["configs"] = cfgs
            if shq is not None:
                line["sharded_quant"] = shq
            line["subrecords_s"] = round(time.perf_counter() - sub_t0, 2)
    mark("subrecords")
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            Tc = min(T, args.cpu_sample_tokens) if args.cpu_sample_tokens > 0 else T
            if gpu_check is not None and Tc < T:  # the same tokens of what the GPU produced (token axis: -2 of the rows, -1 of the scales)
                for key in ("x", "q", "out"):
                    for name in ("k", "v"):
                        gpu_check[f"{key}_{name}"] = np.ascontiguousarray(gpu_check[f"{key}_{name}"][..., :Tc, :])
                for name in ("k", "v"):
                    gpu_check[f"s_{name}"] = np.ascontiguousarray(gpu_check[f"s_{name}"][..., :Tc])
            line["cpu_baseline"] = cpu_baseline(L, B, H, Tc, D, args.cpu_sample_layers, reps=max(1, args.cpu_reps), gpu_check=gpu_check)
            if Tc < T:
                line["cpu_baseline"]["tokens_sampled"] = f"first {Tc} of {T} tokens (per-token work: the rates do not depend on T)"
            mark("cpu_baseline")
        line["device_state"] = device_state
        line["run_s"] = round(time.perf_counter() - t_run0, 2)
        phases["startup_before_workload"] = round(t_run0 - _T_PROCESS0, 2)
        line["phases_s"] = phases
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
