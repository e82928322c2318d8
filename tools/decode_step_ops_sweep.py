#!/usr/bin/env python3
"""Round 4: the per-decode-step launches of the cache path at T = 1 (and a few tokens), by batch size — is any of them as far
from its bytes as the abs-max phase was (21 us for 8.4 MB)? Llama-3-8B layer set [32, B, 8, Tcap, 128]: quantise-append of
the new token(s) (INT8 and INT4), dequantise-append into the fp16 staging buffer, each as ONE launch over the 32 layers.
HIP events bound to each launch's dispatch; bytes = algorithmic (3.0 / 2.5 B per element)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import _kernels_of, _time_launches  # noqa: E402


def main():
    from efficient_llm_inference_amd import kernels as K
    dev = torch.device("cuda:0")
    L, H, D, Tcap = 32, 8, 128, 1032
    for B in (1, 8, 64):
        for T in (1, 4):
            x = torch.randn(L, B, H, T, D, device=dev, dtype=torch.float16)
            ws = torch.empty(L * T, device=dev, dtype=torch.float32)
            row = {"B": B, "T": T, "elements": L * B * H * T * D}
            for kind in ("int8", "int4"):
                Dq = K.packed_dim(kind, D)
                store = torch.zeros(L, B, H, Tcap, Dq, device=dev, dtype=K.QDTYPE[kind])
                scales = torch.zeros(L, Tcap, device=dev, dtype=torch.float32)
                staging = torch.zeros(L, B, H, Tcap, D, device=dev, dtype=torch.float16)
                t0 = 700
                qwin, swin, owin = store[:, :, :, t0:t0 + T], scales[:, t0:t0 + T], staging[:, :, :, t0:t0 + T]
                qf = lambda i: K.quant_tokens(x, qwin, swin, ws, kind)  # noqa: E731
                df = lambda i: K.dequant_tokens(qwin, swin, owin, kind)  # noqa: E731
                for name, fn, bpe in (("quant", qf, 3.0 if kind == "int8" else 2.5), ("dequant", df, 3.0 if kind == "int8" else 2.5)):
                    kern = _kernels_of(lambda: fn(0))
                    ms = _time_launches(fn, 30, warm=3)
                    us = sum(ms) / len(ms) * 1e3
                    row[f"{name}_{kind}"] = {"us": round(us, 2), "GBps": round(row["elements"] * bpe / us / 1e3, 1), "kernel": kern.split("(")[0][:48]}
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
