#!/usr/bin/env python3
"""Round 4: which kernel does each entry point pick, and how far from its bytes is it, over the shapes a server actually
sees — batch 1 / 8 / 64 x head_dim 64 / 128 x fp16 / bf16, prefill chunks of 512 and 2,048 tokens? One row per shape:
quantise INT8 / INT4 and dequantise INT8 / INT4 (kernel, us, fraction of 8 TB/s on 3.0 / 2.5 B per element). The BASELINE
configurations are in bench.py; this is the matrix around them (looking for paths that fall off a fast kernel)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import _kernels_of, _time_launches  # noqa: E402

SHAPES = [  # name, (L, B, H, T, D), dtype
    ("llama3_8b b1 T2048", (32, 1, 8, 2048, 128), torch.float16),
    ("llama3_8b b8 T2048", (32, 8, 8, 2048, 128), torch.float16),
    ("llama3_8b b64 T512", (32, 64, 8, 512, 128), torch.float16),
    ("llama3_8b b8 T2048 bf16", (32, 8, 8, 2048, 128), torch.bfloat16),
    ("llama2_7b b8 T1024 (32 kv heads)", (32, 8, 32, 1024, 128), torch.float16),
    ("llama32_1b b8 T2048 (head_dim 64)", (16, 8, 8, 2048, 64), torch.float16),
    ("llama32_1b b64 T512 (head_dim 64)", (16, 64, 8, 512, 64), torch.float16),
    ("gpt2 b1 T1024", (12, 1, 12, 1024, 64), torch.float16),
    ("gpt2 b8 T1024", (12, 8, 12, 1024, 64), torch.float16),
    ("gpt2 b64 T512", (12, 64, 12, 512, 64), torch.float16),
    ("gpt2-medium b8 T1024", (24, 8, 16, 1024, 64), torch.float16),
    ("head_dim 256 b1 T2048", (16, 1, 4, 2048, 256), torch.float16),
    ("head_dim 96 b1 T2048", (16, 1, 8, 2048, 96), torch.float16),
]


def main():
    from efficient_llm_inference_amd import _lib
    from efficient_llm_inference_amd import kernels as K
    for kv in sys.argv[1:]:  # KEY=VALUE tunables (e.g. quant_wide_min=8192)
        k, v = kv.split("=")
        _lib.set_tunable(k, int(v))
    dev = torch.device("cuda:0")
    for name, (L, B, H, T, D), dt in SHAPES:
        n = L * B * H * T * D
        x = [torch.randn(L, B, H, T, D, device=dev, dtype=dt) for _ in range(2)]
        row = {"shape": name, "LBHTD": [L, B, H, T, D], "MB_in": round(n * 2 / 1e6, 1)}
        for kind in ("int8", "int4"):
            Dq = K.packed_dim(kind, D)
            q = [torch.empty(L, B, H, T, Dq, device=dev, dtype=K.QDTYPE[kind]) for _ in range(2)]
            sc = torch.empty(L, T, device=dev, dtype=torch.float32)
            ws = torch.empty(L * T, device=dev, dtype=torch.float32)
            out = [torch.empty(L, B, H, T, D, device=dev, dtype=dt) for _ in range(2)]
            bpe = 3.0 if kind == "int8" else 2.5
            for op, fn in (("quant", lambda i: K.quant_tokens(x[i & 1], q[i & 1], sc, ws, kind)),
                           ("dequant", lambda i: K.dequant_tokens(q[i & 1], sc, out[i & 1], kind))):
                kern = _kernels_of(lambda: fn(0))
                ms = _time_launches(fn, 12, warm=2)
                us = sum(ms) / len(ms) * 1e3
                row[f"{op}_{kind}"] = {"kernel": " + ".join(k.split("(")[0][:46] for k in kern.split(" + ")), "us": round(us, 1), "frac": round(n * bpe / (us * 1e-6) / 8e12, 3)}
            del q, out
        print(json.dumps(row), flush=True)
        del x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
