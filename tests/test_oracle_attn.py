"""Pins oracle.decode_attention (scope row N1, second form) on the CPU: the reference has no
attention code of its own — `to_past_key_values` (ops.py:345-355) hands dequantised tensors to
the HF model — so the oracle is checked against torch's scaled_dot_product_attention (what HF
runs) over the oracle's own dequantised values plus the exact new token."""
import numpy as np
import pytest
import torch

from oracle import kvq_oracle as O


@pytest.mark.parametrize("B,Hq,Hkv,T,D", [(1, 4, 4, 37, 64), (2, 8, 2, 130, 32), (1, 6, 3, 1, 16), (1, 2, 2, 0, 8)])
@pytest.mark.parametrize("kinds", [("int8", "int4"), ("int4", "int8")])
@pytest.mark.parametrize("with_new", [True, False])
def test_decode_attention_oracle_vs_torch_sdpa(B, Hq, Hkv, T, D, kinds, with_new):
    if T == 0 and not with_new:
        pytest.skip("nothing to attend to")
    rng = np.random.default_rng(B * 1000 + T)
    k = rng.standard_normal((1, B, Hkv, T, D)).astype(np.float16)
    v = rng.standard_normal((1, B, Hkv, T, D)).astype(np.float16)
    kq, _, ks = O.quantize_tokens(k, kinds[0])
    vq, _, vs = O.quantize_tokens(v, kinds[1])
    q = rng.standard_normal((B, Hq, D)).astype(np.float16)
    kn = rng.standard_normal((B, Hkv, D)).astype(np.float16) if with_new else None
    vn = rng.standard_normal((B, Hkv, D)).astype(np.float16) if with_new else None
    sm = 1.0 / np.sqrt(D)
    got = O.decode_attention(q, kq[0], ks[0], kinds[0], vq[0], vs[0], kinds[1], D, sm, kn, vn)

    kd = torch.from_numpy(O.dequantize_tokens(kq, ks, kinds[0], D, "f16")[0]).double()
    vd = torch.from_numpy(O.dequantize_tokens(vq, vs, kinds[1], D, "f16")[0]).double()
    if with_new:
        kd = torch.cat([kd, torch.from_numpy(kn).double()[:, :, None]], dim=2)
        vd = torch.cat([vd, torch.from_numpy(vn).double()[:, :, None]], dim=2)
    rep = Hq // Hkv
    ref = torch.nn.functional.scaled_dot_product_attention(
        torch.from_numpy(q).double()[:, :, None], kd.repeat_interleave(rep, dim=1), vd.repeat_interleave(rep, dim=1),
        scale=sm)[:, :, 0].numpy()
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-11)
