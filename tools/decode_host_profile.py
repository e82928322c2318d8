#!/usr/bin/env python3
"""Where does the host time of one decode step go? cProfile of KVCacheBenchmarker.generate_with_quantized_kv (staged path,
the reference's loop benchmarker.py:422-491) beside the full-cache loop, gpt2 random-init, prompt 512 + N new tokens.
  python tools/decode_host_profile.py [method] [new_tokens]  -> cumulative-time table restricted to this package + totals"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import efficient_llm_inference_amd as E  # noqa: E402
from efficient_llm_inference_amd.benchmarking.offline import load_model  # noqa: E402

method = sys.argv[1] if len(sys.argv) > 1 else "quant_int8"
n_new = int(sys.argv[2]) if len(sys.argv) > 2 else 256
model, tok = load_model("gpt2", "cuda", torch.float16)
bm = E.KVCacheBenchmarker(model, tok, device="cuda")
for m in (method, "full_cache"):
    bm.benchmark_method(["<64>"], method=m, max_new_tokens=8)
torch.cuda.synchronize()
for m in ("full_cache", method):
    t0 = time.perf_counter()
    r = bm.benchmark_method(["<512>"], method=m, max_new_tokens=n_new)
    torch.cuda.synchronize()
    print(f"{m}: {r['tokens_per_sec']:.1f} tok/s, {(time.perf_counter() - t0) / n_new * 1e3:.3f} ms per token (wall, incl. prefill)")
pr = cProfile.Profile()
pr.enable()
bm.benchmark_method(["<512>"], method=method, max_new_tokens=n_new)
torch.cuda.synchronize()
pr.disable()
for sort, pat in (("cumulative", "efficient|kvq"), ("tottime", None)):
    s = io.StringIO()
    ps = pstats.Stats(pr, stream=s).sort_stats(sort)
    ps.print_stats(*( [pat, 45] if pat else [30]))
    print(s.getvalue()[:9000])
