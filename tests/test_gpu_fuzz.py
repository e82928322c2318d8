"""Seeded fuzz over shapes, dtypes, strides and windows: every entry point vs the oracle on ~100
small random configurations (odd D, D % 8 == 0 but not a power of two, B > 1, tiny T, guard
tokens around store windows). Deterministic; run on the MI355X with ``-m gpu``."""
import numpy as np
import pytest
import torch

from oracle import kvq_oracle as O
from tests.util import TD, bits, odt, seeded_kv, to_numpy, to_torch

pytestmark = pytest.mark.gpu

D_CHOICES = [1, 2, 5, 8, 16, 24, 32, 40, 64, 72, 96, 128, 256]


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        yield (int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(1, 9)), int(rng.integers(1, 40)),
               int(rng.choice(D_CHOICES)), str(rng.choice(["f16", "bf16", "f32"])), str(rng.choice(["int8", "int4"])),
               str(rng.choice(["normal", "heavy", "tiny"])), int(rng.integers(0, 4)), int(rng.integers(0, 2**31)))


def test_fuzz_quant_dequant():
    from efficient_llm_inference_amd import kernels
    for (G, B, H, T, D, dtype, kind, dist, pad, seed) in _cases(120, 1234):
        shape = (G, B, H, T, D)
        x_np = seeded_kv(shape, dtype, seed, dist)
        q_ref, _, s32_ref = O.quantize_tokens(x_np, kind, dtype=odt(dtype))
        x = to_torch(x_np, dtype)
        Dq = kernels.packed_dim(kind, D)
        store = torch.zeros(G, B, H, T + pad + 1, Dq, dtype=kernels.QDTYPE[kind], device="cuda")
        scales = torch.zeros(G, T + pad + 1, dtype=torch.float32, device="cuda")
        ws = torch.empty(G * T + 4, dtype=torch.float32, device="cuda")
        src = [x[g] for g in range(G)] if seed % 2 else x
        kernels.quant_tokens(src, store[:, :, :, pad:pad + T], scales[:, pad:pad + T], ws, kind)
        tag = (shape, dtype, kind, dist, pad)
        assert np.array_equal(to_numpy(store[:, :, :, pad:pad + T]), q_ref), tag
        assert np.array_equal(bits(scales[:, pad:pad + T]), bits(s32_ref)), tag
        assert int(store[:, :, :, :pad].to(torch.int32).abs().sum()) == 0 and int(store[:, :, :, pad + T:].to(torch.int32).abs().sum()) == 0, tag
        od = ["f16", "bf16", "f32"][seed % 3]
        out = torch.zeros(G, B, H, T + pad, D, dtype=TD[od], device="cuda")
        kernels.dequant_tokens(store[:, :, :, pad:pad + T], scales[:, pad:pad + T], out[:, :, :, pad:], kind)
        assert np.array_equal(bits(out[:, :, :, pad:]), bits(O.dequantize_tokens(q_ref, s32_ref, kind, D, od))), tag


def test_fuzz_eviction():
    import efficient_llm_inference_amd as E
    from efficient_llm_inference_amd import cache as C
    for (G, B, H, T, D, dtype, _kind, dist, pad, seed) in _cases(80, 99):
        G = 2 * ((G + 1) // 2)
        shape = (G, B, H, T, D)
        x_np = seeded_kv(shape, dtype, seed, dist)
        big = torch.zeros(G, B, H, T + 2 * pad, D, dtype=TD[dtype], device="cuda")
        big[:, :, :, pad:pad + T] = to_torch(x_np, dtype)
        past = tuple((big[2 * l, :, :, pad:pad + T], big[2 * l + 1, :, :, pad:pad + T]) for l in range(G // 2))
        rng = np.random.default_rng(seed)
        chunk, keep, W = int(rng.integers(1, 9)), int(rng.integers(0, 12)), int(rng.integers(1, 20))
        P, stride = int(rng.integers(0, 5)), int(rng.integers(1, 5))
        tag = (shape, dtype, chunk, keep, W, P, stride)
        res = E.chunk_summarize_kv(past, chunk, keep)
        orc = O.chunk_summarize_kv(x_np, chunk, keep, dtype=odt(dtype))
        win = E.trim_kv_sliding_window(past, W)
        stz = C.trim_kv_strided(past, W, stride, P)
        bud = C.trim_kv_budget_old(past, W, 3, P)
        blk = C.trim_kv_block_old(past, W, 4, 2, P)
        idx_s = O.keep_indices_strided(T, W, stride, P)
        idx_b = O.keep_indices_budget_old(T, W, 3, P)
        idx_k = O.keep_indices_block_old(T, W, 4, 2, P)
        for l in range(G // 2):
            for j in (0, 1):
                g = 2 * l + j
                assert np.array_equal(bits(res[l][j]), bits(orc[g])), tag
                assert np.array_equal(bits(win[l][j]), bits(O.trim_kv_sliding_window(x_np[g], W))), tag
                assert np.array_equal(bits(stz[l][j]), bits(O.gather_tokens(x_np[g], idx_s))), tag
                assert np.array_equal(bits(bud[l][j]), bits(O.gather_tokens(x_np[g], idx_b))), tag
                assert np.array_equal(bits(blk[l][j]), bits(O.gather_tokens(x_np[g], idx_k))), tag
