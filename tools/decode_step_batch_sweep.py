#!/usr/bin/env python3
"""Round 4: kvq_decode_step (attention over the store + the new token's quantise-append, fused in the merge launch) against
kvq_decode_attn (attention only) by batch size — what does the fused quantise-append cost once the new token's [B,H,1,D]
slice no longer fits the one-round-trip register path (more than 8,192 elements: batch > 8 at 8 kv heads x 128)?
Llama-3-8B layer shape, 2,048 stored tokens, INT8 K + INT4 V; wall time per call over 200 calls (one stream, no sync inside)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from efficient_llm_inference_amd import _lib
    from efficient_llm_inference_amd import kernels as K
    dev = torch.device("cuda:0")
    Hq, Hkv, D, T, cap = 32, 8, 128, 2048, 2056
    only = os.environ.get("KVQ_SWEEP_ONLY")  # e.g. "64:decode_step" (one batch size, one call: for a rocprofv3 run)
    for B in ((int(only.split(":")[0]),) if only else (1, 8, 16, 64)):
        q = torch.randn(B, Hq, D, device=dev, dtype=torch.float16)
        kn = torch.randn(B, Hkv, D, device=dev, dtype=torch.float16)
        vn = torch.randn(B, Hkv, D, device=dev, dtype=torch.float16)
        ks = torch.randint(-127, 127, (B, Hkv, cap, D), device=dev, dtype=torch.int8)
        vs = torch.randint(0, 255, (B, Hkv, cap, D // 2), device=dev, dtype=torch.uint8)
        ksc = torch.rand(cap, device=dev) * 0.01 + 0.001
        vsc = torch.rand(cap, device=dev) * 0.01 + 0.001
        out = torch.empty_like(q)
        ws = torch.empty(K.decode_attn_workspace(B, Hq, Hkv, cap, D), device=dev, dtype=torch.float32)
        plan = K.DecodeStepPlan(q, ks, ksc, "int8", vs, vsc, "int4", 1e-8)
        sm = D ** -0.5
        row = {"B": B, "new_token_elements": B * Hkv * D}
        for name, fn in (("decode_attn", lambda: K.decode_attn(q, ks, ksc, "int8", vs, vsc, "int4", T, out, ws, sm, kn, vn)),
                         ("decode_step", lambda: K.decode_step(plan, q, kn, vn, T, out, ws, sm))):
            if only and only.split(":")[1] != name:
                continue
            _lib.kernel_log_clear()
            fn()
            torch.cuda.synchronize()
            kern = [k.split("(")[0][:44] for k in _lib.kernel_log()]
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 200
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            row[name] = {"us_per_call": round((time.perf_counter() - t0) / n * 1e6, 2), "kernels": kern}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
