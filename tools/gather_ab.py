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


def patterns():
    """`--patterns`: WHY is block_old (8 of every 64 tokens) slower than stride 4? The same number of 2 KiB runs (8 consecutive
    token rows), the same output, only WHERE in its 16 KiB block each run sits:
      last8      block_old's own list: every run at offset 14 KiB of its block (address bits 11-13 always 111)
      first8     every run at offset 0
      rotate     run k at offset (k mod 8) * 2 KiB: the runs cover every residue below 16 KiB equally
      rows_4     32-token blocks, 4 kept (1 KiB runs, 8 KiB spacing), last-4 and rotated
    If the policy's fixed offset is what costs, `rotate` runs at the stride-4 rate and first8 = last8."""
    from efficient_llm_inference_amd import kernels as K
    dev = torch.device("cuda:0")
    L, B, H, T, D = 32, 8, 8, 32768, 128
    torch.manual_seed(42)
    xs = [torch.randn(B, H, T, D, device=dev, dtype=torch.float16) for _ in range(2 * L)]

    def runs(block, keep, where):
        idx = []
        for k, start in enumerate(range(0, T, block)):
            off = {"last": block - keep, "first": 0, "rotate": (k * keep) % block}[where]
            idx.extend(range(start + off, start + off + keep))
        return idx

    cases = {"last8_of_64": runs(64, 8, "last"), "first8_of_64": runs(64, 8, "first"), "rotate8_of_64": runs(64, 8, "rotate"),
             "last4_of_32": runs(32, 4, "last"), "rotate4_of_32": runs(32, 4, "rotate"),
             "last16_of_128": runs(128, 16, "last"), "rotate16_of_128": runs(128, 16, "rotate"),
             "stride_8": list(range(0, T, 8)), "stride_4": list(range(0, T, 4))}
    for rnd in range(3):
        for name, keep in cases.items():
            idx = torch.tensor(keep, dtype=torch.int32, device=dev)
            out = torch.empty(2 * L, B, H, len(keep), D, dtype=torch.float16, device=dev)
            ms = _time_launches(lambda i: K.gather_tokens(xs, out, idx), 5, warm=1)
            avg = sum(ms) / len(ms)
            print(json.dumps({"round": rnd, "pattern": name, "kept": len(keep), "us": round(avg * 1e3, 1),
                              "frac": round(4.0 * 2 * L * B * H * len(keep) * D / (avg * 1e-3) / 8e12, 4)}), flush=True)
            del out
            torch.cuda.empty_cache()


def main():
    if "--patterns" in sys.argv:
        return patterns()
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
