#!/usr/bin/env python3
"""Round 4: the two row-gather kernels against each other in ONE process, alternating (`gather_rows` 1 = gather_rows_k, 4 KiB
items; 0 = gather_tokens_k, grid-stride), on the config-5 per-GPU share (64 tensors of [8,8,32768,128] fp16) for every
index-select policy at the reference's default arguments. Each cell: HIP events bound to the launch's own dispatch."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import _time_launches  # noqa: E402


def main():
    from efficient_llm_inference_amd import _lib
    from efficient_llm_inference_amd import cache as C
    dev = torch.device("cuda:0")
    L, B, H, T, D = 32, 8, 8, 32768, 128
    torch.manual_seed(42)
    past = tuple((torch.randn(B, H, T, D, device=dev, dtype=torch.float16), torch.randn(B, H, T, D, device=dev, dtype=torch.float16)) for _ in range(L))
    calls = {
        "trim_kv_strided": lambda: C.trim_kv_strided(past, window_size=256, stride=4, prefix_len=32),
        "trim_kv_block_old": lambda: C.trim_kv_block_old(past, window_size=256, block_size=64, keep_per_block=8, prefix_len=32),
        "trim_kv_budget_old": lambda: C.trim_kv_budget_old(past, window_size=256, old_budget=64, prefix_len=32),
        "trim_kv_prefix_window": lambda: C.trim_kv_prefix_window(past, prefix_len=32, window_size=256),
        "trim_kv_strided_2": lambda: C.trim_kv_strided(past, window_size=256, stride=2, prefix_len=32),
    }
    for rnd in range(3):
        for name, fn in calls.items():
            row = {"round": rnd, "op": name}
            for which in (1, 0):
                _lib.set_tunable("gather_rows", which)
                kept = fn()[0][0].size(2)
                ms = _time_launches(lambda i: fn(), 5, warm=1)
                avg = sum(ms) / len(ms)
                row[f"gather_rows_{which}"] = {"us": round(avg * 1e3, 1), "frac": round(4.0 * 2 * L * B * H * kept * D / (avg * 1e-3) / 8e12, 4)}
                torch.cuda.empty_cache()
            print(json.dumps(row), flush=True)
    _lib.set_tunable("gather_rows", 1)


if __name__ == "__main__":
    main()
