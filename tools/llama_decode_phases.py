#!/usr/bin/env python3
"""Where a decode step of the full Llama-3-8B architecture (random init, 16K-token prompt) spends its time, per
cache path: wall time per token after the prompt, for full_cache (prompt truncated to 1024 tokens, as the reference
does) / fullctx (fp16 cache, whole prompt) / quant_<mode> staged / fused / graph; `phases` splits the staged loop.

    python tools/llama_decode_phases.py [path ...] [--prompt 16000] [--new 64] [--mode mixed] [--arch llama-8b]

Under rocprofv3 (`rocprofv3 --kernel-trace --stats -- python3 tools/llama_decode_phases.py staged`) the kernel
table shows what the GPU ran for that path.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="*", default=["full", "staged", "fused", "graph"])
    ap.add_argument("--prompt", type=int, default=16000)
    ap.add_argument("--new", type=int, default=64)
    ap.add_argument("--mode", default="mixed")
    ap.add_argument("--arch", default="llama-8b")
    args = ap.parse_args()
    import efficient_llm_inference_amd as E
    from efficient_llm_inference_amd.benchmarking.offline import load_model
    model, tok = load_model(args.arch, "cuda", torch.float16)
    bm = E.KVCacheBenchmarker(model, tok, device="cuda")
    prompt = f"<{args.prompt}>"

    def run(path, n_new):
        bm.fused_attention = path in ("fused", "graph")
        bm.graph_decode = path == "graph"
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if path == "full":  # the reference's full_cache: prompt truncated to 1024 tokens (benchmarker.py:113)
            bm.generate_with_cache(prompt, n_new)
        elif path == "fullctx":  # the fp16 cache at the quantised paths' context length (no truncation)
            enc = bm._encode
            bm._encode = lambda p, truncate: enc(p, False)
            try:
                bm.generate_with_cache(prompt, n_new)
            finally:
                bm._encode = enc
        else:
            bm.generate_with_quantized_kv(prompt, n_new, mode=args.mode)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def staged_phases(n_new):
        """The in-place staged loop of generate_with_quantized_kv with a device synchronisation after each phase."""
        from efficient_llm_inference_amd.benchmarking.benchmarker import to_legacy_tuple
        from efficient_llm_inference_amd.quantization import QuantizedKVCache, hf_cache
        ids = tok(prompt, return_tensors="pt").input_ids.to("cuda")
        with torch.no_grad():
            out = model(input_ids=ids, use_cache=True)
            logits = out.logits[:, -1, :]
            past = to_legacy_tuple(out.past_key_values)
            qc = QuantizedKVCache(n_layers=len(past), mode=args.mode, device="cuda", compute_dtype=torch.float16)
            qc.reserve(ids.shape[-1] + n_new)
            qc.init_from_prompt_past(past)
            del out, past
            staged = hf_cache.StagedQuantizedCache(qc)
            acc = {"sync": 0.0, "forward": 0.0, "commit": 0.0}
            for i in range(n_new):
                nxt = torch.argmax(logits, dim=-1, keepdim=True)
                for name, fn in (("sync", lambda: staged.sync()), ("forward", None), ("commit", lambda: staged.commit())):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    if name == "forward":
                        out = model(input_ids=nxt, use_cache=True, past_key_values=staged.cache)
                        logits = out.logits[:, -1, :]
                    else:
                        fn()
                    torch.cuda.synchronize()
                    if i >= 2:
                        acc[name] += time.perf_counter() - t0
        print("staged phases, ms per token: " + "  ".join(f"{k} {v / (n_new - 2) * 1e3:.3f}" for k, v in acc.items()), flush=True)

    if "phases" in args.paths:
        staged_phases(args.new)
        args.paths.remove("phases")
    for path in args.paths:
        run(path, 4)  # warm-up: lazy initialisations, allocator
        short = run(path, 8)
        long = run(path, 8 + args.new)
        longer = run(path, 8 + 4 * args.new)
        print(f"{path:<7} prompt+8 tokens {short * 1e3:9.1f} ms   per further token {(long - short) / args.new * 1e3:8.3f} ms"
              f" (next {args.new}) / {(longer - long) / (3 * args.new) * 1e3:8.3f} ms (next {3 * args.new})"
              f"   ({3 * args.new / (longer - long):7.1f} tok/s after the prompt)", flush=True)


if __name__ == "__main__":
    main()
