"""Pins the oracle's index-list restatement of the sparse eviction family (scope row N3) to the
kept-token indices read back from the reference's own outputs (tests/golden/g7_sparse.npz). CPU."""
import numpy as np
import pytest

from oracle import kvq_oracle as O

TS = (5, 40, 41, 100, 300, 1000)
WP = ((8, 0), (8, 4), (32, 3), (256, 32))


@pytest.fixture(scope="module")
def g7():
    from tests.conftest import load_golden
    return load_golden("g7_sparse.npz")


def _expect(idx, T):
    return list(range(T)) if idx is None else idx


@pytest.mark.parametrize("T", TS)
@pytest.mark.parametrize("W,P", WP)
def test_index_lists_match_reference(g7, T, W, P):
    assert _expect(O.keep_indices_prefix_window(T, P, W), T) == g7[f"prefix.T{T}.W{W}.P{P}"].tolist()
    for stride in (1, 3, 4):
        assert _expect(O.keep_indices_strided(T, W, stride, P), T) == g7[f"strided.T{T}.W{W}.P{P}.s{stride}"].tolist()
    for bs, kpb in ((16, 4), (64, 8), (7, 7)):
        assert _expect(O.keep_indices_block_old(T, W, bs, kpb, P), T) == g7[f"block.T{T}.W{W}.P{P}.b{bs}.k{kpb}"].tolist()
    for budget in (0, 1, 5, 64):
        assert _expect(O.keep_indices_budget_old(T, W, budget, P), T) == g7[f"budget.T{T}.W{W}.P{P}.n{budget}"].tolist()


def test_paged_golden_is_identity(g7):
    kv = g7["paged.kv"]
    assert np.array_equal(g7["paged.k"], kv[0]) and np.array_equal(g7["paged.v"], kv[1])
    assert g7["paged.meta"].tolist() == [3, 2 * 3 * 2 * 3 * 8 * 8 * 2, 21 * 2 * 3 * 8 * 2 * 2]


def test_round2_edges_match_reference():
    """tests/golden/g8_round2.npz: window_size == 0 (the reference's `-0:` slice keeps the whole tensor,
    prefix+window then repeats the prefix) and budget index lists at lengths with non-integral spacing."""
    from tests.conftest import load_golden
    g8 = load_golden("g8_round2.npz")
    for T in (1, 7):
        x = np.arange(T, dtype=np.float32)[:, None] * np.ones((1, 2), np.float32)
        assert O.trim_kv_sliding_window(x, 0)[:, 0].astype(np.int64).tolist() == g8[f"win0.T{T}"].tolist() == list(range(T))
        for P in (0, 3):
            assert _expect(O.keep_indices_prefix_window(T, P, 0), T) == g8[f"prefix0.T{T}.P{P}"].tolist()
    for T in (97, 513, 2049, 4096, 16385, 32768):
        for (W, P, n) in ((8, 0, 7), (256, 32, 64), (33, 5, 100), (1, 1, 3)):
            assert _expect(O.keep_indices_budget_old(T, W, n, P), T) == g8[f"budget.T{T}.W{W}.P{P}.n{n}"].tolist(), (T, W, P, n)
