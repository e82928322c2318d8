#!/usr/bin/env python3
"""Round 4, VERDICT item 1: does the quantise tile kernel (quant_tile_k, rows a1 / a2) care about its row strides?

Config 4's K / V sets are [32, 1, 8, 16384, 128] fp16: the 8 head rows a one-wave tile reads are 4 MiB apart, the rows it
writes 2 MiB (INT8) / 1 MiB (INT4) apart — all powers of two. This script times the SAME kernel on the same bytes with
the rows padded: the input as a [:, :, :, :T] view of a [G,B,H,T+pad_in,D] allocation, the store as a [:, :, :, :T] view
of [G,B,H,T+pad_out,Dq] (what `_KVStore.reserve` would allocate with a padded Tcap). pad = 16 tokens makes the input row
stride 4 MiB + 4 KiB (an odd multiple of 4 KiB), 17 tokens an odd multiple of 256 B, and so on.

  python tools/quant_stride_sweep.py                 # the whole table (time per pad pair, both kinds, N(0,1) input)
  python tools/quant_stride_sweep.py --dist outlier  # SURVEY §8d's second distribution (1 % of channels x 8)
  python tools/quant_stride_sweep.py --one 16,16     # ONE pad pair, a few launches: the command the PMC passes wrap

Timing: HIP events bound to each launch's own dispatch (bench._time_launches); two rotating input sets and stores.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import _kernels_of, _time_launches  # noqa: E402

L, B, H, T, D = 32, 1, 8, 16384, 128
BPE = {"int8": 3.0, "int4": 2.5}


def make_input(pad, dist, seed, dev):
    torch.manual_seed(seed)
    full = torch.randn(L, B, H, T + pad, D, device=dev, dtype=torch.float16)
    if dist == "outlier":  # 1 % of the (head, d) channels carry 8 x the magnitude in every token (typical of keys)
        g = torch.Generator(device="cpu").manual_seed(seed)
        mask = (torch.rand(H, D, generator=g) < 0.01).to(dev)
        full = torch.where(mask[None, None, :, None, :], full * 8, full)
    return full[:, :, :, :T]


def make_store(kind, pad, dev):
    Dq = D if kind == "int8" else D // 2
    q = torch.empty(L, B, H, T + pad, Dq, device=dev, dtype=torch.int8 if kind == "int8" else torch.uint8)
    return q[:, :, :, :T], torch.zeros(L, T, device=dev, dtype=torch.float32)


def measure(pad_in, pad_out, dist, iters, dev, check=False):
    from efficient_llm_inference_amd import kernels as K
    n_in = max(2, -(-(768 << 20) // (L * B * H * T * D * 2)))  # inputs rotate over >= 768 MiB
    xs = [make_input(pad_in, dist, 42 + i, dev) for i in range(n_in)]
    ws = torch.empty(L * T, device=dev, dtype=torch.float32)
    rec = {"pad_in_tokens": pad_in, "pad_out_tokens": pad_out, "in_row_stride_B": (T + pad_in) * D * 2}
    for kind in ("int8", "int4"):
        n_st = max(2, -(-(512 << 20) // (L * B * H * T * D // (1 if kind == "int8" else 2))))
        stores = [make_store(kind, pad_out, dev) for _ in range(n_st)]
        fn = lambda i: K.quant_tokens(xs[i % n_in], stores[i % n_st][0], stores[i % n_st][1], ws, kind)  # noqa: E731
        kern = _kernels_of(lambda: fn(0))
        ms = _time_launches(fn, iters)
        nbytes = L * B * H * T * D * BPE[kind]
        avg = sum(ms) / len(ms)
        rec["shape_LBHTD"] = [L, B, H, T, D]
        rec[kind] = {"kernel": kern.split("(")[0][-60:], "avg_us": round(avg * 1e3, 2), "median_us": round(ms[len(ms) // 2] * 1e3, 2),
                     "min_us": round(ms[0] * 1e3, 2), "frac_of_8TBps": round(nbytes / (avg * 1e-3) / 8e12, 4),
                     "out_row_stride_B": (T + pad_out) * (D if kind == "int8" else D // 2)}
        if check and pad_in + pad_out:  # same bytes as the unpadded call: the layout must not change a single result
            q0, s0 = make_store(kind, 0, dev)
            x0 = xs[0].contiguous()
            K.quant_tokens(x0, q0, s0, ws, kind)
            fn(0)
            torch.cuda.synchronize()
            rec[kind]["equal_to_unpadded"] = bool(torch.equal(q0, stores[0][0]) and torch.equal(s0, stores[0][1]))
        del stores
    del xs
    torch.cuda.empty_cache()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dist", default="normal", choices=("normal", "outlier"))
    ap.add_argument("--iters", type=int, default=24)
    ap.add_argument("--one", default=None, help="pad_in,pad_out: measure only this pair (for the PMC passes)")
    ap.add_argument("--pads", default="0,1,4,16,17,48,80,272,1040")
    ap.add_argument("--shape", default=None, help="L,B,H,T,D instead of config 4's 32,1,8,16384,128")
    args = ap.parse_args()
    if args.shape:
        global L, B, H, T, D
        L, B, H, T, D = (int(v) for v in args.shape.split(","))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    if args.one:
        pi, po = (int(v) for v in args.one.split(","))
        print(json.dumps({"dist": args.dist, **measure(pi, po, args.dist, args.iters, dev)}), flush=True)
        return
    pads = [int(v) for v in args.pads.split(",")]
    pairs = [(0, 0)] + [(p, 0) for p in pads if p] + [(0, p) for p in pads if p] + [(p, p) for p in pads if p] + [(0, 0)]
    for pi, po in pairs:
        print(json.dumps({"dist": args.dist, **measure(pi, po, args.dist, args.iters, dev, check=True)}), flush=True)


if __name__ == "__main__":
    main()
