"""A transformers Cache whose layers live in the quantised cache's persistent staging buffers.

The reference rebuilds a ``DynamicCache`` from a freshly dequantised tuple before every forward
(reference src/benchmarking/benchmarker.py:470-471) and the model then ``torch.cat``s one token onto
every layer: O(T) bytes per step twice over. Here the model's cache layers ARE the staging buffers
of :class:`QuantizedKVCache` (``[L, B, H, Tcap, D]``, capacity reserved up front):

  * before a forward, ``sync()`` dequantises only the tokens appended since the last call;
  * during the forward, ``update()`` writes the new token's exact K/V into slot ``T`` in place and
    returns views ``[.., :T+1, :]`` — no ``cat``, no reallocation;
  * after the forward, ``commit()`` quantises slot ``T`` into the store and dequantises it back
    into the same slot, so the next forward sees exactly what the reference's loop would see.

Per decode step the cache policy therefore moves O(1) tokens instead of O(T). Values handed to the
model are bit-identical to the tuple path (tests/test_gpu_benchmarker.py).

Needs transformers >= 4.54 / 5.x (``DynamicLayer``); ``available()`` says whether it can be used.
"""
from __future__ import annotations

import torch

try:
    from transformers.cache_utils import DynamicCache, DynamicLayer
except Exception:  # pragma: no cover - older transformers: the tuple path is used instead
    DynamicCache = None
    DynamicLayer = None


def available() -> bool:
    return DynamicLayer is not None


if DynamicLayer is not None:

    class _StagedLayer(DynamicLayer):
        """One layer's K/V as windows of the staging buffers; ``update`` appends in place."""

        def __init__(self, owner: "StagedQuantizedCache", index: int):
            super().__init__()
            self._owner = owner
            self._index = index
            self.is_initialized = True
            self._rebind()

        def _rebind(self) -> None:
            o, i = self._owner, self._index
            self._kbuf, self._vbuf = o._qc._k.stage[i], o._qc._v.stage[i]  # [B,H,cap,D]
            self.dtype, self.device = self._kbuf.dtype, self._kbuf.device
            self._len = o._len
            self.keys = self._kbuf[:, :, :self._len]
            self.values = self._vbuf[:, :, :self._len]

        def update(self, key_states: torch.Tensor, value_states: torch.Tensor, *args, **kwargs):
            n = key_states.shape[-2]
            if self._len + n > self._kbuf.shape[2]:
                self._owner._grow(self._len + n)  # rare: capacity was not reserved
            self._kbuf[:, :, self._len:self._len + n] = key_states
            self._vbuf[:, :, self._len:self._len + n] = value_states
            self._len += n
            self.keys = self._kbuf[:, :, :self._len]
            self.values = self._vbuf[:, :, :self._len]
            return self.keys, self.values

        def get_seq_length(self) -> int:
            return self._len


class StagedQuantizedCache:
    """Binds a :class:`QuantizedKVCache` to a transformers ``DynamicCache`` of in-place layers."""

    def __init__(self, qcache):
        if not available():
            raise RuntimeError("kvq: this transformers version has no DynamicLayer; use the tuple path")
        self._qc = qcache
        self._len = 0
        self.cache = None  # the object handed to model(..., past_key_values=...)

    def _grow(self, need: int) -> None:
        self._qc.reserve(max(need, 2 * self._qc._k.cap))
        for layer in self.cache.layers:
            n = layer._len
            layer._rebind()
            layer._len = n
            layer.keys, layer.values = layer._kbuf[:, :, :n], layer._vbuf[:, :, :n]

    def sync(self):
        """Make the staging buffers current (dequantise what is new) and return the HF cache."""
        qc = self._qc
        qc._k.dequant_staged(qc.compute_dtype)
        qc._v.dequant_staged(qc.compute_dtype)
        self._len = qc._k.lens[0]
        if self.cache is None:
            self.cache = DynamicCache()
            self.cache.layers = [_StagedLayer(self, i) for i in range(len(qc.layers))]
        else:
            moved = self.cache.layers[0]._kbuf.data_ptr() != qc._k.stage[0].data_ptr()  # store grew
            for layer in self.cache.layers:
                if moved:
                    layer._rebind()
                layer._len = self._len  # the views handed to attention are rebuilt by update()
        return self.cache

    def commit(self) -> None:
        """After a forward: quantise the tokens the model appended in place (slots [T, T+n) of the
        staging buffers) into the stores; the next ``sync`` replaces them by their dequantised
        values. Equivalent to ``append_from_past`` on the model's returned cache
        (reference ops.py:323-330)."""
        qc = self._qc
        new_len = self.cache.layers[0]._len
        T = qc._k.lens[0]
        if new_len == T:
            return
        qc._k.append(qc._k.stage[:, :, :, T:new_len])  # one 5-D window: all layers, one launch
        qc._v.append(qc._v.stage[:, :, :, T:new_len])
