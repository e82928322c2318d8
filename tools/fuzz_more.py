#!/usr/bin/env python3
"""One-off widening of tests/test_gpu_fuzz.py: the same two fuzz loops (every entry point against the oracle on small random
shapes, dtypes, strides and windows) under other seeds than the suite's fixed pair, with larger T and more groups mixed in.
  python tools/fuzz_more.py [n_seeds]   -> one line per seed; a mismatch raises with the failing configuration"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import tests.test_gpu_fuzz as F  # noqa: E402


def cases_with(seed_shift, big):
    def _cases(n, seed):
        rng = np.random.default_rng(seed + seed_shift)
        for _ in range(n):
            T = int(rng.integers(1, 40)) if not big else int(rng.choice([1, 2, 63, 64, 65, 127, 129, 255, 257, 300]))
            G = int(rng.integers(1, 4)) if not big else int(rng.choice([1, 2, 5]))
            yield (G, int(rng.integers(1, 4)), int(rng.integers(1, 9)), T, int(rng.choice(F.D_CHOICES)),
                   str(rng.choice(["f16", "bf16", "f32"])), str(rng.choice(["int8", "int4"])),
                   str(rng.choice(["normal", "heavy", "tiny"])), int(rng.integers(0, 4)), int(rng.integers(0, 2**31)))
    return _cases


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    for s in range(1, n + 1):
        t0 = time.perf_counter()
        F._cases = cases_with(1000 * s, big=(s % 3 == 0))
        F.test_fuzz_quant_dequant()
        F.test_fuzz_eviction()
        print(f"seed shift {1000 * s} ({'larger T' if s % 3 == 0 else 'suite ranges'}): 200 configurations ok in {time.perf_counter() - t0:.1f} s", flush=True)


if __name__ == "__main__":
    main()
