#!/usr/bin/env python3
"""Offline counterpart of the reference's examples/quantized_cache.py: full cache vs quant_int8 /
quant_int4 / quant_mixed on the MI355X, with tokens/sec, estimated KV-cache MB and the text
similarity of each quantised generation to the full-cache one.

    python examples/quantized_cache_offline.py [gpt2 | gpt2-medium | /path/to/local/checkpoint]

Without a local checkpoint the model is RANDOM-INIT (no network here), so the similarity column
only shows how far quantisation noise moves a greedy decode of an untrained model.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from efficient_llm_inference_amd import Config, KVCacheBenchmarker  # noqa: E402
from efficient_llm_inference_amd.benchmarking.offline import load_model  # noqa: E402
from efficient_llm_inference_amd.evaluation import text_similarity  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "gpt2"
    cfg = Config(model_name=name)
    if cfg.device != "cuda":
        raise SystemExit("the quantised cache path needs the MI355X (no CPU fallback)")
    model, tok = load_model(name, cfg.device, cfg.dtype, seed=cfg.seed)
    bench = KVCacheBenchmarker(model, tok, device=cfg.device)
    prompts = ["<128>", "<256>", "<384>"] if not os.path.isdir(name) else [
        "The future of artificial intelligence is", "Efficient inference for large language models requires"]
    ref_text, _ = bench.generate_with_cache(prompts[0], cfg.max_new_tokens)
    print(f"{'method':<18} {'tok/s':>9} {'KV MB':>9} {'similarity':>11}")
    for method, fused, graph in (("full_cache", False, False), ("quant_int8", False, False), ("quant_mixed", False, False),
                                 ("quant_int4", False, False), ("quant_int8", True, False), ("quant_mixed", True, False),
                                 ("quant_int4", True, False), ("quant_int8", True, True), ("quant_mixed", True, True),
                                 ("quant_int4", True, True)):
        # fused: the model attends straight over the INT8 / INT4 store (no fp16 copy of the cache)
        # graph: the fused decode step captured once into a HIP graph and replayed per token
        bench.fused_attention = fused
        bench.graph_decode = graph
        res = bench.benchmark_method(prompts, method=method, max_new_tokens=cfg.max_new_tokens)
        if method == "full_cache":
            sim = 1.0
        else:
            text, _, _ = bench.generate_with_quantized_kv(prompts[0], cfg.max_new_tokens, mode=method[6:])
            sim = text_similarity(ref_text, text)
        label = method + ("+graph" if graph else "+fused" if fused else "")
        print(f"{label:<18} {res['tokens_per_sec']:>9.1f} {res['est_kv_cache_mb_avg']:>9.3f} {sim:>11.3f}")
    bench.fused_attention = False
    bench.graph_decode = False


if __name__ == "__main__":
    main()
