#!/usr/bin/env python3
"""Round 4: kvq_decode_step_dev (device-side token count: what a captured decode graph replays; one-tile splits) against
kvq_decode_step (host-side count; the LDS-staged ring kernel on large batches) — what does the graph path give up at
serving batch sizes? Llama-3-8B layer shape, INT8 K + INT4 V; wall per call over 200 calls."""
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
    Hq, Hkv, D = 32, 8, 128
    for B, T in ((1, 16384), (8, 2048), (8, 16384), (64, 2048)):
        cap = T + 8
        q = torch.randn(B, Hq, D, device=dev, dtype=torch.float16)
        kn = torch.randn(B, Hkv, D, device=dev, dtype=torch.float16)
        vn = torch.randn(B, Hkv, D, device=dev, dtype=torch.float16)
        ks = torch.randint(-127, 127, (B, Hkv, cap, D), device=dev, dtype=torch.int8)
        vs = torch.randint(0, 255, (B, Hkv, cap, D // 2), device=dev, dtype=torch.uint8)
        ksc = torch.rand(cap, device=dev) * 0.01 + 0.001
        vsc = torch.rand(cap, device=dev) * 0.01 + 0.001
        out = torch.empty_like(q)
        ws = torch.empty(K.decode_attn_workspace_cap(B, Hq, Hkv, cap, D), device=dev, dtype=torch.float32)
        plan = K.DecodeStepPlan(q, ks, ksc, "int8", vs, vsc, "int4", 1e-8)
        t_dev = torch.full((1,), T, dtype=torch.int32, device=dev)
        sm = D ** -0.5
        row = {"B": B, "T": T, "MB": round(B * Hkv * T * (D + D // 2) / 1e6, 1)}
        for name, fn in (("decode_step (host T)", lambda: K.decode_step(plan, q, kn, vn, T, out, ws, sm)),
                         ("decode_step_dev (device T, bound T+4)", lambda: K.decode_step_dev(plan, q, kn, vn, t_dev, T + 4, out, ws, sm))):
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
            us = (time.perf_counter() - t0) / n * 1e6
            row[name] = {"us_per_call": round(us, 2), "TBps": round(row["MB"] / us, 3), "kernels": kern}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
