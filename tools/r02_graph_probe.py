"""Feasibility probe (GPU box): can one decode step of an HF model whose attention is kvq_fused be
captured into a HIP graph (torch.cuda.CUDAGraph) and replayed? Prints eager vs replay step times.
The captured step still carries a host-side T (frozen), so replays recompute the same position —
this probe only answers 'does capture work and what does a replay cost'."""
import sys
import time

import torch

sys.path.insert(0, ".")
import efficient_llm_inference_amd as E  # noqa: E402
from efficient_llm_inference_amd.benchmarking.offline import load_model  # noqa: E402
from efficient_llm_inference_amd.quantization import fused_attention as FA  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "gpt2"
model, tok = load_model(arch, "cuda", torch.float16)
ids = tok("<512>", return_tensors="pt").input_ids.cuda()
cfg = model.config
L = getattr(cfg, "num_hidden_layers", None) or cfg.n_layer
fc = FA.FusedQuantizedCache(L, mode="int8", device="cuda", compute_dtype=torch.float16, reserve=ids.shape[-1] + 600)
with torch.no_grad(), FA.fused_attention(model, fc) as cache:
    out = model(input_ids=ids, use_cache=True, past_key_values=cache)
    nxt = out.logits[:, -1].argmax(-1, keepdim=True)
    for _ in range(5):
        out = model(input_ids=nxt, use_cache=True, past_key_values=cache)
        nxt = out.logits[:, -1].argmax(-1, keepdim=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        out = model(input_ids=nxt, use_cache=True, past_key_values=cache)
        nxt = out.logits[:, -1].argmax(-1, keepdim=True)
    torch.cuda.synchronize()
    print(f"eager fused step: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
    T = fc.qcache._k.lens[0]
    static_ids = nxt.clone()
    cache_pos = torch.tensor([T], device="cuda")
    fc.ignore_decode_mask = True
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(s):
            for _ in range(2):
                out = model(input_ids=static_ids, use_cache=True, past_key_values=cache, cache_position=cache_pos)
        torch.cuda.current_stream().wait_stream(s)
        print("warm-up with explicit cache_position ok; lens", fc.qcache._k.lens[0])
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = model(input_ids=static_ids, use_cache=True, past_key_values=cache, cache_position=cache_pos)
            static_next = out.logits[:, -1].argmax(-1, keepdim=True)
        print("capture ok")
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize()
        print(f"graph replay step: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms  next={int(static_next)}")
    except Exception as exc:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        print("CAPTURE FAILED:", repr(exc)[:500])
