"""Scope row N3 on the MI355X: the index-select eviction family and the paged layout, vs the
reference's goldens (kept-token indices read back from identifiable rows) and vs the oracle."""
import numpy as np
import pytest
import torch

from oracle import kvq_oracle as O
from tests.conftest import load_golden
from tests.util import bits, odt, seeded_kv, to_numpy, to_torch

pytestmark = pytest.mark.gpu

TS = (5, 40, 41, 100, 300, 1000)
WP = ((8, 0), (8, 4), (32, 3), (256, 32))


@pytest.fixture(scope="module")
def E():
    assert torch.cuda.is_available()
    import efficient_llm_inference_amd as pkg
    from efficient_llm_inference_amd import _lib
    _lib.load()
    return pkg


@pytest.fixture(scope="module")
def g7():
    return load_golden("g7_sparse.npz")


def _ident(T):
    x = torch.arange(T, dtype=torch.float32)[None, None, :, None] + torch.arange(8)[None, None, None, :] / 16.0
    return x.expand(1, 2, T, 8).contiguous().half().cuda()


def _kept(k):
    return k[0, 0, :, 0].float().round().long().cpu().tolist()


@pytest.mark.parametrize("T", TS)
@pytest.mark.parametrize("W,P", WP)
def test_golden_kept_indices(E, g7, T, W, P):
    from efficient_llm_inference_amd import cache as C
    x = _ident(T)
    past = ((x, x),)
    (k, v), = C.trim_kv_prefix_window(past, prefix_len=P, window_size=W)
    assert _kept(k) == g7[f"prefix.T{T}.W{W}.P{P}"].tolist() and torch.equal(k, v)
    if T <= P + W:
        assert k is x  # unchanged object, as the reference
    for stride in (1, 3, 4):
        (k, _), = C.trim_kv_strided(past, window_size=W, stride=stride, prefix_len=P)
        assert _kept(k) == g7[f"strided.T{T}.W{W}.P{P}.s{stride}"].tolist()
    for bs, kpb in ((16, 4), (64, 8), (7, 7)):
        (k, _), = C.trim_kv_block_old(past, window_size=W, block_size=bs, keep_per_block=kpb, prefix_len=P)
        assert _kept(k) == g7[f"block.T{T}.W{W}.P{P}.b{bs}.k{kpb}"].tolist()
    for budget in (0, 1, 5, 64):
        (k, _), = C.trim_kv_budget_old(past, window_size=W, old_budget=budget, prefix_len=P)
        assert _kept(k) == g7[f"budget.T{T}.W{W}.P{P}.n{budget}"].tolist()


@pytest.mark.parametrize("shape", [(6, 1, 8, 700, 128), (4, 2, 3, 90, 5), (2, 1, 4, 260, 24)])
@pytest.mark.parametrize("dtype", ["f16", "bf16", "f32"])
def test_oracle_gather_multi_layer(E, shape, dtype):
    """2L tensors per launch; vector (16 B) and scalar (odd D) kernels; strided inputs."""
    from efficient_llm_inference_amd import cache as C
    G, B, H, T, D = shape
    x_np = seeded_kv(shape, dtype, 17, "normal")
    x = to_torch(x_np, dtype)
    past = tuple((x[2 * l], x[2 * l + 1]) for l in range(G // 2))
    cases = [
        (C.trim_kv_prefix_window(past, 5, 40), O.keep_indices_prefix_window(T, 5, 40)),
        (C.trim_kv_strided(past, 33, 3, 2), O.keep_indices_strided(T, 33, 3, 2)),
        (C.trim_kv_block_old(past, 20, 16, 5, 1), O.keep_indices_block_old(T, 20, 16, 5, 1)),
        (C.trim_kv_budget_old(past, 17, 9, 3), O.keep_indices_budget_old(T, 17, 9, 3)),
    ]
    for res, idx in cases:
        for l in range(G // 2):
            assert np.array_equal(bits(res[l][0]), bits(O.gather_tokens(x_np[2 * l], idx)))
            assert np.array_equal(bits(res[l][1]), bits(O.gather_tokens(x_np[2 * l + 1], idx)))
    big = torch.zeros(G, B, H, T + 6, D, dtype=x.dtype, device="cuda")
    big[:, :, :, 3:T + 3] = x
    past = tuple((big[2 * l, :, :, 3:T + 3], big[2 * l + 1, :, :, 3:T + 3]) for l in range(G // 2))
    res = C.trim_kv_strided(past, 33, 3, 2)
    idx = O.keep_indices_strided(T, 33, 3, 2)
    for l in range(G // 2):
        assert np.array_equal(bits(res[l][0]), bits(O.gather_tokens(x_np[2 * l], idx)))


def test_paged_cache(E, g7):
    from efficient_llm_inference_amd.cache import PagedKVCache
    kv = to_torch(g7["paged.kv"])  # [2,B,H,21,D]
    pc = PagedKVCache(block_size=8, device="cuda", dtype=torch.float16)
    with pytest.raises(ValueError, match="Empty cache"):
        pc.get_kv()
    for t in range(21):
        pc.append(kv[0, :, :, t:t + 1], kv[1, :, :, t:t + 1])
    k, v = pc.get_kv()
    assert np.array_equal(to_numpy(k), g7["paged.k"]) and np.array_equal(to_numpy(v), g7["paged.v"])
    assert [pc.num_blocks(), pc.allocated_bytes(), pc.used_bytes()] == g7["paged.meta"].tolist()
    assert len(pc.k_blocks) == 3 and pc.k_blocks[0].shape == (2, 3, 8, 8)
    pc2 = PagedKVCache(block_size=8, device="cuda", dtype=torch.float16)
    pc2.extend(kv[0, :, :, :13], kv[1, :, :, :13])
    pc2.extend(kv[0, :, :, 13:], kv[1, :, :, 13:])
    k2, v2 = pc2.get_kv()
    assert torch.equal(k, k2) and torch.equal(v, v2) and pc2.num_blocks() == 3
    # exact multiples of the block size and growth of the block pool
    x = torch.randn(1, 4, 200, 64, device="cuda").half()
    pc3 = PagedKVCache(block_size=16, device="cuda", dtype=torch.float16)
    pc3.extend(x[:, :, :192], x[:, :, :192] * 2)
    k3, v3 = pc3.get_kv()
    assert torch.equal(k3, x[:, :, :192]) and pc3.num_blocks() == 12
    for t in range(192, 200):
        pc3.append(x[:, :, t:t + 1], x[:, :, t:t + 1] * 2)
    k3, v3 = pc3.get_kv()
    assert torch.equal(k3, x) and torch.equal(v3, x * 2) and pc3.num_blocks() == 13


def test_all_twelve_methods_on_gpu(E):
    from efficient_llm_inference_amd import KVCacheBenchmarker
    from efficient_llm_inference_amd.benchmarking.benchmarker import VALID_METHODS
    from efficient_llm_inference_amd.benchmarking.offline import load_model
    model, tok = load_model("gpt2-tiny", "cuda", torch.float16)
    b = KVCacheBenchmarker(model, tok, device="cuda")
    for method in VALID_METHODS:
        res = b.benchmark_method(["<70>"], method=method, max_new_tokens=5, window_size=16, block_size=8, chunk_size=8,
                                 keep_last=16, prefix_len=4, stride=3, keep_per_block=2, old_budget=6)
        assert res["method"] == method and 1 <= res["total_new_tokens"] <= 5
    # the paged layout is lossless: same tokens as the model's own full cache
    t_full, _ = b.generate_with_cache("<40>", 8)
    t_paged, n, alloc_mb, used_mb, nb = b.generate_with_paged_attention("<40>", 8, block_size=16)
    assert t_paged == t_full and n == 8 and nb == 2 * 3 and alloc_mb >= used_mb > 0


def test_gather_out_of_range_index_is_not_dereferenced(E):
    """kvq_gather_tokens takes a DEVICE index table; an index outside [0, T) (stale table,
    negative value) must not become an address. The kernel writes a row of zeros there; in-range
    rows are the exact gather."""
    from efficient_llm_inference_amd import kernels as K
    G, B, H, T, D = 2, 1, 3, 10, 64
    x = torch.randn(G, B, H, T, D, device="cuda", dtype=torch.float16)
    idx = torch.tensor([0, 9, 10, -1, 2**31 - 1, 4], dtype=torch.int32, device="cuda")
    out = torch.full((G, B, H, idx.numel(), D), 7.0, device="cuda", dtype=torch.float16)
    K.gather_tokens(x, out, idx)
    torch.cuda.synchronize()
    good = torch.tensor([0, 9, 4], device="cuda")
    assert torch.equal(out[:, :, :, [0, 1, 5]], x[:, :, :, good])
    assert float(out[:, :, :, 2:5].abs().max()) == 0.0
    # the 2-byte-granular path (rows that are not 16-byte multiples)
    x2 = torch.randn(1, 1, 2, 5, 3, device="cuda", dtype=torch.float16)
    idx2 = torch.tensor([4, 5, 1], dtype=torch.int32, device="cuda")
    out2 = torch.full((1, 1, 2, 3, 3), 7.0, device="cuda", dtype=torch.float16)
    K.gather_tokens(x2, out2, idx2)
    torch.cuda.synchronize()
    assert torch.equal(out2[:, :, :, 0], x2[:, :, :, 4]) and torch.equal(out2[:, :, :, 2], x2[:, :, :, 1])
    assert float(out2[:, :, :, 1].abs().max()) == 0.0


def test_round2_edges_on_gpu(E):
    """g8_round2.npz through the HIP path: window_size == 0 keeps everything (sliding) / repeats the
    prefix (prefix+window), and the budget policy's device-evaluated fp32 linspace gives the index lists
    the reference's CPU run produced."""
    from efficient_llm_inference_amd.cache import trim_kv_budget_old, trim_kv_prefix_window
    g8 = load_golden("g8_round2.npz")
    for T in (1, 7):
        x = _ident(T)
        (k, v), = E.trim_kv_sliding_window(((x, x),), 0)
        assert k is x and v is x and _kept(k) == g8[f"win0.T{T}"].tolist()
        for P in (0, 3):
            (k, _), = trim_kv_prefix_window(((x, x),), prefix_len=P, window_size=0)
            assert _kept(k) == g8[f"prefix0.T{T}.P{P}"].tolist(), (T, P)
    for T in (97, 513, 2049, 4096, 16385, 32768):
        x = torch.arange(T, dtype=torch.float32, device="cuda")[None, None, :, None].expand(1, 1, T, 2).contiguous()
        for (W, P, n) in ((8, 0, 7), (256, 32, 64), (33, 5, 100), (1, 1, 3)):
            (k, _), = trim_kv_budget_old(((x, x),), window_size=W, old_budget=n, prefix_len=P)
            assert k[0, 0, :, 0].round().long().cpu().tolist() == g8[f"budget.T{T}.W{W}.P{P}.n{n}"].tolist(), (T, W, P, n)


@pytest.mark.parametrize("shape", [(6, 1, 8, 700, 128), (3, 2, 12, 333, 64), (2, 8, 8, 2048, 128), (1, 1, 2, 50, 24)])
def test_gather_4kib_items_equal_the_grid_stride_kernel(E, shape):
    """gather_rows (round 4, default 1): gather_rows_k — one 16-byte piece per thread, 4 KiB of output per workgroup, the
    (batch row, head) as a grid dimension — against the grid-stride kernel it replaced (gather_rows = 0): equal bytes for every
    policy's index list, out-of-range indices (zero rows), a strided source, the legacy tuple of separate tensors; head_dim 24
    (48-byte rows: 3 pieces per row, not a power of two) stays on the grid-stride kernel either way."""
    from efficient_llm_inference_amd import _lib
    from efficient_llm_inference_amd import kernels as K
    G, B, H, T, D = shape
    g = torch.Generator(device="cuda").manual_seed(T)
    big = torch.randn(G, B, H, T + 5, D, device="cuda", generator=g).half()
    x = big[:, :, :, 2:T + 2]
    lists = [O.keep_indices_strided(T, 33, 3, 2), O.keep_indices_block_old(T, 20, 16, 5, 1), O.keep_indices_budget_old(T, 17, 9, 3),
             [0, T - 1, T, -1, 5, 5, 2**31 - 1, 1]]
    old = _lib.get_tunable("gather_rows")
    try:
        for keep in lists:
            idx = torch.tensor(list(keep), dtype=torch.int32, device="cuda")
            outs = []
            for which in (1, 0):
                _lib.set_tunable("gather_rows", which)
                for src in (x, [x[i] for i in range(G)]):
                    out = torch.full((G, B, H, idx.numel(), D), 7.0, device="cuda", dtype=torch.float16)
                    _lib.kernel_log_clear()
                    K.gather_tokens(src, out, idx)
                    torch.cuda.synchronize()
                    name = _lib.kernel_log()[0]
                    assert name.startswith("gather_rows_k" if which == 1 and D % 8 == 0 and (D // 8) & (D // 8 - 1) == 0 else "gather_tokens_k<"), (which, name)
                    outs.append(out)
            for o in outs[1:]:
                assert torch.equal(o.view(torch.int16), outs[0].view(torch.int16))
            ok = [j for j, t in enumerate(keep) if 0 <= t < T]
            assert torch.equal(outs[0][:, :, :, ok], x[:, :, :, [keep[j] for j in ok]])
    finally:
        _lib.set_tunable("gather_rows", old)


def test_paged_cache_with_more_blocks_than_one_pointer_table(E):
    """A pool of more than 128 blocks (the pointer table one launch carries) is ONE allocation: kvq_window_compact stitches it in
    one launch, the kernel adding block * stride itself (round 4). 250 full blocks + a ragged one, K and V, against the input."""
    from efficient_llm_inference_amd import _lib
    from efficient_llm_inference_amd.cache import PagedKVCache
    x = torch.randn(2, 3, 1003, 64, device="cuda").half()
    pc = PagedKVCache(block_size=4, device="cuda", dtype=torch.float16)
    pc.extend(x, x * 2)
    assert pc.num_blocks() == 251
    _lib.kernel_log_clear()
    k, v = pc.get_kv()
    torch.cuda.synchronize()
    assert _lib.kernel_log() == ["copy_rows_k<true>"]
    assert torch.equal(k, x) and torch.equal(v, x * 2)
    # the pointer-list form (separately allocated tensors) still goes 128 per launch: 130 tensors, window of 5 tokens
    from efficient_llm_inference_amd import kernels as K
    xs = [torch.randn(1, 2, 9, 8, device="cuda").half() for _ in range(130)]
    out = torch.empty(130, 1, 2, 5, 8, device="cuda", dtype=torch.float16)
    K.window_compact(xs, out, 5)
    torch.cuda.synchronize()
    assert all(torch.equal(out[i], xs[i][:, :, -5:]) for i in range(130))
