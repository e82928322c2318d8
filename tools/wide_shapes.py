#!/usr/bin/env python3
"""Batched-slice quantise by batch size: GB/s of single-pass bytes for [L, B, 8, T, 128] fp16 slices, INT8 and INT4,
the default route (quant_wide_k for 16384 < B*H*D <= 131072) beside quant_wide = 0 (split phases).
usage: python tools/wide_shapes.py [L]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import efficient_llm_inference_amd as E  # noqa: E402,F401
from efficient_llm_inference_amd import _lib, kernels as K  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda", 0)
    for B, T in ((32, 1024), (64, 512), (128, 256), (48, 512), (64, 1)):
        H, D = 8, 128
        x = [torch.randn(L, B, H, T, D, device=dev, dtype=torch.float16) for _ in range(2)]
        for kind, bpe in (("int8", 3.0), ("int4", 2.5)):
            Dq = K.packed_dim(kind, D)
            q = [torch.empty(L, B, H, T, Dq, device=dev, dtype=K.QDTYPE[kind]) for _ in range(2)]
            sc = torch.empty(L, T, device=dev, dtype=torch.float32)
            ws = torch.empty(L * T, device=dev, dtype=torch.float32)
            for wide in (1, 0):
                _lib.set_tunable("quant_wide", wide)
                try:
                    _lib.kernel_log_clear()
                    K.quant_tokens(x[0], q[0], sc, ws, kind)
                    names = " + ".join(_lib.kernel_log())
                    for i in range(3):
                        K.quant_tokens(x[i & 1], q[i & 1], sc, ws, kind)
                    torch.cuda.synchronize()
                    n = 10 if T > 1 else 200
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(n):
                        K.quant_tokens(x[i & 1], q[i & 1], sc, ws, kind)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / n
                    print(f"B={B:4d} T={T:5d} {kind} wide={wide}  {ms * 1e3:9.1f} us  {L * B * H * T * D * bpe / ms / 1e6:8.1f} GB/s  {names}", flush=True)
                finally:
                    _lib.set_tunable("quant_wide", 1)
        del x, q


if __name__ == "__main__":
    main()
