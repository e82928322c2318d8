"""KV eviction on MI355X over HIP: sliding window, chunk summary, the index-select family
(prefix+window, strided, block, budget) and the simulated paged cache.

Same names, arguments and return structure as the reference
(reference src/cache/implementations.py:124-140 and :295-346): they take and return the legacy
tuple ``tuple_L[(k, v)]`` of ``[B, H, T, D]`` tensors.

MI355X-first differences (results identical):
  * all 2L tensors of a call are processed by ONE kernel launch (their base pointers travel in
    the kernel-argument segment) into ONE output buffer; the returned tuple holds views of it.
  * ``trim_kv_sliding_window`` materialises the window (the reference returns views and lets the
    next ``torch.cat`` move the bytes); when ``T <= window_size`` the input objects are returned
    untouched, exactly like the reference (:135).
  * ``chunk_summarize_kv`` is a single pass (the reference zero-pads with ``torch.cat``, calls
    ``mean`` and concatenates again: three passes, :326-343).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from .. import _lib, kernels


def _flatten(past_key_values) -> List[torch.Tensor]:
    flat: List[torch.Tensor] = []
    for k, v in past_key_values:
        flat.append(k)
        flat.append(v)
    return flat


def _by_signature(tensors: List[torch.Tensor]) -> Dict[tuple, List[int]]:
    """indices grouped by (shape, strides, dtype, device): one launch per group"""
    groups: Dict[tuple, List[int]] = {}
    for i, t in enumerate(tensors):
        if t.dim() != 4:
            raise ValueError(f"kvq: KV tensors must be [B,H,T,D], got {tuple(t.shape)}")
        _lib.require_gpu(t, "past_key_values")
        groups.setdefault((tuple(t.shape), tuple(t.stride()), t.dtype, t.device), []).append(i)
    return groups


def _prep(t: torch.Tensor) -> torch.Tensor:
    return t if (t.size(-1) == 1 or t.stride(-1) == 1) else t.contiguous()


def _pairs(past_key_values, flat: List[torch.Tensor], out: List[torch.Tensor]) -> tuple:
    """Re-assemble ``tuple_L[(k, v)]``; a tensor the policy left untouched is returned as the
    caller's own object (the reference returns its inputs unchanged in that case)."""
    orig = _flatten(past_key_values)
    return tuple((orig[2 * l] if out[2 * l] is flat[2 * l] else out[2 * l],
                  orig[2 * l + 1] if out[2 * l + 1] is flat[2 * l + 1] else out[2 * l + 1])
                 for l in range(len(out) // 2))


def trim_kv_sliding_window(past_key_values: tuple, window_size: int) -> tuple:
    """Keep only the last ``window_size`` tokens of every K and V (reference
    implementations.py:124-140)."""
    flat = [_prep(t) for t in _flatten(past_key_values)]
    out: List[torch.Tensor] = list(flat)
    for (shape, _strides, dtype, device), idx in _by_signature(flat).items():
        B, H, T, D = shape
        if not T > window_size or window_size == 0:
            # unchanged objects, as the reference; window_size == 0 keeps everything there too
            # (`k[:, :, -0:, :]` is the whole tensor, reference implementations.py:137-139)
            continue
        W = int(window_size)
        for c0 in range(0, len(idx), 256):
            part = idx[c0:c0 + 256]
            buf = torch.empty(len(part), B, H, W, D, dtype=dtype, device=device)
            kernels.window_compact([flat[i] for i in part], buf, W)
            for j, i in enumerate(part):
                out[i] = buf[j]
    return _pairs(past_key_values, flat, out)


def chunk_summarize_kv(past_key_values: tuple, chunk_size: int, keep_last: int) -> tuple:
    """Replace tokens older than the last ``keep_last`` by mean-pooled summaries of ``chunk_size``
    tokens each; the ragged last chunk is zero-padded, i.e. still divided by ``chunk_size``
    (reference implementations.py:295-346)."""
    flat = [_prep(t) for t in _flatten(past_key_values)]
    out: List[torch.Tensor] = list(flat)
    for (shape, _strides, dtype, device), idx in _by_signature(flat).items():
        B, H, T, D = shape
        keep = min(int(keep_last), T)
        if T - keep <= 0:
            continue  # nothing to compress: unchanged objects (reference :316-318)
        Tout = kernels.chunk_summary_len(T, chunk_size, keep_last)
        for c0 in range(0, len(idx), 256):
            part = idx[c0:c0 + 256]
            buf = torch.empty(len(part), B, H, Tout, D, dtype=dtype, device=device)
            kernels.chunk_meanpool([flat[i] for i in part], buf, int(chunk_size), int(keep_last))
            for j, i in enumerate(part):
                out[i] = buf[j]
    return _pairs(past_key_values, flat, out)


# ----------------------------------------------------------------------------- index-select family
# (scope row N3) Each policy keeps [prefix] + [selected older tokens] + [dense tail]; the kept
# token indices are host integers (they depend only on T and the policy), the byte movement is ONE
# row-gather launch over all 2L tensors (kvq_gather_tokens) instead of index_select + cat per tensor.


def _gather_policy(past_key_values: tuple, index_fn) -> tuple:
    flat = [_prep(t) for t in _flatten(past_key_values)]
    out: List[torch.Tensor] = list(flat)
    for (shape, _strides, dtype, device), idx in _by_signature(flat).items():
        B, H, T, D = shape
        keep = index_fn(T, device)
        if keep is None:
            continue  # short sequence: unchanged objects, as the reference
        if keep and (min(keep) < 0 or max(keep) >= T):
            raise ValueError("kvq: eviction policy produced an out-of-range token index")
        idx_dev = torch.tensor(keep, dtype=torch.int32).to(device, non_blocking=True)
        for c0 in range(0, len(idx), 256):
            part = idx[c0:c0 + 256]
            buf = torch.empty(len(part), B, H, len(keep), D, dtype=dtype, device=device)
            kernels.gather_tokens([flat[i] for i in part], buf, idx_dev)
            for j, i in enumerate(part):
                out[i] = buf[j]
    return _pairs(past_key_values, flat, out)


def trim_kv_prefix_window(past_key_values, prefix_len: int, window_size: int):
    """Keep the first ``prefix_len`` and the last ``window_size`` tokens (reference
    implementations.py:143-154)."""
    def index_fn(T, _device):
        if T <= prefix_len + window_size:
            return None
        # window_size == 0: the reference's `k[:, :, -0:, :]` is the WHOLE tensor (:151-152), so the result is
        # the prefix followed by every token
        return list(range(prefix_len)) + (list(range(T)) if window_size == 0 else list(range(T - window_size, T)))
    return _gather_policy(past_key_values, index_fn)


def trim_kv_strided(past_key_values, window_size: int, stride: int, prefix_len: int = 0):
    """Keep the prefix, every ``stride``-th older token and the dense tail (reference
    implementations.py:157-190)."""
    assert stride >= 1

    def index_fn(T, _device):
        if T <= prefix_len + window_size:
            return None
        tail_start = max(prefix_len, T - window_size)
        return list(range(prefix_len)) + list(range(prefix_len, tail_start, stride)) + list(range(tail_start, T))
    return _gather_policy(past_key_values, index_fn)


def trim_kv_block_old(past_key_values, window_size: int, block_size: int = 64, keep_per_block: int = 8,
                      prefix_len: int = 0):
    """Keep the prefix, the last ``keep_per_block`` tokens of every ``block_size`` block of the
    older region, and the dense tail (reference implementations.py:193-245)."""
    assert block_size >= 1
    assert 1 <= keep_per_block <= block_size

    def index_fn(T, _device):
        if T <= prefix_len + window_size:
            return None
        tail_start = max(prefix_len, T - window_size)
        old, start = [], prefix_len
        while start < tail_start:
            end = min(start + block_size, tail_start)
            old.extend(range(max(start, end - keep_per_block), end))
            start = end
        return list(range(prefix_len)) + old + list(range(tail_start, T))
    return _gather_policy(past_key_values, index_fn)


def trim_kv_budget_old(past_key_values, window_size: int, old_budget: int = 64, prefix_len: int = 0):
    """Keep the prefix, ``old_budget`` older tokens sampled uniformly (fp32 ``linspace`` truncated
    to integers, consecutive duplicates removed) and the dense tail (reference
    implementations.py:248-292)."""
    assert old_budget >= 0

    def index_fn(T, device):
        if T <= prefix_len + window_size:
            return None
        tail_start = max(prefix_len, T - window_size)
        old_len = tail_start - prefix_len
        old: List[int] = []
        if old_len > 0 and old_budget > 0:
            if old_len <= old_budget:
                old = list(range(prefix_len, tail_start))
            else:
                # the reference's own expression (:279-282) evaluated where the reference evaluates it — on the
                # tensors' device — so that the fp32 linspace truncation is the device kernel's; the <= old_budget
                # indices come back to the host (unique_consecutive synchronises in the reference as well)
                sel = torch.linspace(prefix_len, tail_start - 1, steps=old_budget, device=device).long()
                old = torch.unique_consecutive(sel).tolist()
        return list(range(prefix_len)) + old + list(range(tail_start, T))
    return _gather_policy(past_key_values, index_fn)


# ----------------------------------------------------------------------------- paged layout


class PagedKVCache:
    """Simulated paged KV cache of one layer: K/V live in fixed-size blocks along the sequence
    (reference implementations.py:10-121). Same methods and byte accounting; the blocks are slices
    of one pool tensor ``[n_blocks, B, H, block_size, D]`` that grows by doubling, an append is one
    small device copy, and ``get_kv`` stitches all full blocks with ONE launch per K/V
    (kvq_window_compact over the block axis) plus one for the ragged last block."""

    def __init__(self, block_size: int = 64, device: str = "cuda", dtype: torch.dtype = torch.float16):
        self.block_size = int(block_size)
        self.device = device
        self.dtype = dtype
        self.t_filled = 0
        self._B = self._H = self._D = None
        self._k_pool = self._v_pool = None
        self._n_blocks = 0

    def num_blocks(self) -> int:
        return self._n_blocks

    @property
    def k_blocks(self):
        return [self._k_pool[i] for i in range(self._n_blocks)]

    @property
    def v_blocks(self):
        return [self._v_pool[i] for i in range(self._n_blocks)]

    def _alloc_block(self, B: int, H: int, D: int) -> None:
        cap = 0 if self._k_pool is None else self._k_pool.size(0)
        if self._n_blocks == cap:
            new_cap = max(4, 2 * cap)
            k = torch.empty(new_cap, B, H, self.block_size, D, device=self.device, dtype=self.dtype)
            v = torch.empty_like(k)
            if cap:
                k[:cap] = self._k_pool
                v[:cap] = self._v_pool
            self._k_pool, self._v_pool = k, v
        self._n_blocks += 1

    @torch.no_grad()
    def append(self, k_1tok: torch.Tensor, v_1tok: torch.Tensor) -> None:
        """Append one token's ``[B, H, 1, D]`` key/value (reference :57-80)."""
        assert k_1tok.dim() == 4 and v_1tok.dim() == 4
        B, H, one, D = k_1tok.shape
        assert one == 1
        _lib.require_gpu(k_1tok, "k_1tok")
        if self._B is None:
            self._B, self._H, self._D = B, H, D
        if self._n_blocks == 0 or (self.t_filled % self.block_size) == 0:
            self._alloc_block(B, H, D)
        blk, off = divmod(self.t_filled, self.block_size)
        self._k_pool[blk, :, :, off:off + 1, :] = k_1tok
        self._v_pool[blk, :, :, off:off + 1, :] = v_1tok
        self.t_filled += 1

    @torch.no_grad()
    def extend(self, k: torch.Tensor, v: torch.Tensor) -> None:
        """Append all ``n`` tokens of ``[B, H, n, D]`` tensors: same state as n ``append`` calls
        (the reference's prompt initialisation loop, benchmarker.py:524-526), block-wise copies."""
        assert k.dim() == 4 and v.dim() == 4 and k.shape == v.shape
        B, H, n, D = k.shape
        _lib.require_gpu(k, "k")
        if self._B is None:
            self._B, self._H, self._D = B, H, D
        done = 0
        while done < n:
            if self._n_blocks == 0 or (self.t_filled % self.block_size) == 0:
                self._alloc_block(B, H, D)
            blk, off = divmod(self.t_filled, self.block_size)
            take = min(self.block_size - off, n - done)
            self._k_pool[blk, :, :, off:off + take, :] = k[:, :, done:done + take, :]
            self._v_pool[blk, :, :, off:off + take, :] = v[:, :, done:done + take, :]
            self.t_filled += take
            done += take

    @torch.no_grad()
    def get_kv(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Stitch the blocks back into ``(K, V)`` of shape ``[B, H, T, D]`` (reference :82-106)."""
        if self.t_filled == 0:
            raise ValueError("Empty cache")
        B, H, D, bs, T = self._B, self._H, self._D, self.block_size, self.t_filled
        outs = []
        for pool in (self._k_pool, self._v_pool):
            out = torch.empty(B, H, T, D, device=self.device, dtype=self.dtype)
            full, rem = divmod(T, bs)
            if full:
                # out viewed as [n_full, B, H, bs, D]: block n lands at tokens [n*bs, (n+1)*bs)
                dst = out[:, :, :full * bs].unflatten(2, (full, bs)).permute(2, 0, 1, 3, 4)
                kernels.window_compact(pool[:full], dst, bs)
            if rem:
                kernels.window_compact(pool[full:full + 1, :, :, :rem], out[None, :, :, full * bs:], rem)
            outs.append(out)
        return outs[0], outs[1]

    def allocated_bytes(self) -> int:
        """Bytes of the allocated blocks, unused slots included (reference :108-115)."""
        if self._n_blocks == 0:
            return 0
        return 2 * self._n_blocks * self._B * self._H * self.block_size * self._D * self.dtype.itemsize

    def used_bytes(self) -> int:
        """Bytes of the tokens actually stored (reference :117-121)."""
        if self.t_filled == 0:
            return 0
        return self.t_filled * self._B * self._H * self._D * self.dtype.itemsize * 2
