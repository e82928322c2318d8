#!/usr/bin/env python3
"""Round 4: where does the one-workgroup-per-(group, token) abs-max kernel (absmax_fewtokens_k) stop paying against the tile
walk (quant_tile_k<..., 1>: rows / 8 one-wave workgroups, atomicMax per token)? kvq_absmax_tokens on [64, 64, 8, T, 128] fp16
(the K + V pair of a batch-64 rank share) for T = 1 ... 64, both kernels, HIP events bound to each launch's dispatch."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import _kernels_of, _time_launches  # noqa: E402


def main():
    from efficient_llm_inference_amd import _lib
    from efficient_llm_inference_amd import kernels as K
    dev = torch.device("cuda:0")
    G, B, H, D = 64, 64, 8, 128
    for T in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64):
        xs = [torch.randn(G, B, H, T, D, device=dev, dtype=torch.float16) for _ in range(2)]
        out = torch.empty(G, T, device=dev, dtype=torch.float32)
        row = {"T": T, "MB": round(G * B * H * T * D * 2 / 1e6, 1)}
        tabs = {}
        for name, knob in (("few_tokens", 1 << 20), ("tile_walk", 0)):
            _lib.set_tunable("quant_few_tokens", knob)
            kern = _kernels_of(lambda: K.absmax_tokens(xs[0], out))
            ms = _time_launches(lambda i: K.absmax_tokens(xs[i & 1], out), 20, warm=3)
            K.absmax_tokens(xs[0], out)
            tabs[name] = out.clone()
            row[name] = {"us": round(sum(ms) / len(ms) * 1e3, 2), "min_us": round(ms[0] * 1e3, 2), "kernel": kern.split("(")[0][:40]}
        row["equal"] = bool(torch.equal(tabs["few_tokens"], tabs["tile_walk"]))
        print(json.dumps(row), flush=True)
        del xs
        torch.cuda.empty_cache()
    _lib.set_tunable("quant_few_tokens", 16)


if __name__ == "__main__":
    main()
