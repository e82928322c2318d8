"""INTEGRATION.md's code is EXECUTED here, so the document cannot drift from the ABI.

Level 1: the ctypes stub a reference maintainer pastes over src/cuda/extensions.py (reference extensions.py:116-147) is
extracted from the markdown, exec'd with KVQ_HIP_LIB pointing at the built library, and its two entry points are checked
against the reference-held G1 fixtures (`dq8.f16` / `dq4.f16`: outputs of the imported reference, tests/golden/make_golden.py),
through the reference's own call sites' argument preparation (ops.py:83-87, :112-119).
Level 2: the quantise-all / dequantise-all call sequence is exec'd on the G5 fixture's tensors and compared with the
reference's stored bytes, scales and to_past_key_values() output; the sharded three-step sequence under a 1-rank RCCL group."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests.conftest import ROOT
from tests.util import bits, to_numpy, to_torch

pytestmark = pytest.mark.gpu
DOC = os.path.join(ROOT, "INTEGRATION.md")


def _blocks(section):
    """python code blocks of the `## <section>` part of INTEGRATION.md, in order"""
    text = open(DOC).read()
    start = text.index(f"## {section}")
    nxt = text.find("\n## ", start + 4)
    return re.findall(r"```python\n(.*?)```", text[start:nxt if nxt > 0 else len(text)], flags=re.S)


@pytest.fixture(scope="module")
def stub():
    from efficient_llm_inference_amd import _lib
    _lib.load()
    src = _blocks("Level 1")[0]
    assert "def get_cuda_extension" in src and "kvq_dequant_i4_f16_flat" in src
    old = os.environ.get("KVQ_HIP_LIB")
    os.environ["KVQ_HIP_LIB"] = _lib.LIB_PATH
    try:
        ns = {"__name__": "src.cuda.extensions"}
        exec(compile(src, "INTEGRATION.md[Level 1]", "exec"), ns)
    finally:
        if old is None:
            del os.environ["KVQ_HIP_LIB"]
        else:
            os.environ["KVQ_HIP_LIB"] = old
    return ns


def test_level1_stub_against_the_reference_fixtures(stub, g1):
    ext = stub["get_cuda_extension"]()
    assert ext is not None and stub["get_cuda_extension"]() is ext  # the module-global singleton of extensions.py:144-147
    n = 0
    for fam in sorted({k.rsplit(".", 1)[0] for k in g1.files if k.endswith(".q8")}):
        name, dt, dist = fam.split(".")
        if dt == "bf16":  # the plugin path is taken for fp16 outputs; a bf16 slice's scale is bf16 (same float() conversion, covered by f16 / f32)
            continue
        q8, p4 = to_torch(g1[fam + ".q8"]), to_torch(g1[fam + ".p4"])
        s8, s4 = float(g1[fam + ".s8"][0]), float(g1[fam + ".s4"][0])  # float(scale): ops.py:87 / :117
        last = int(g1[fam + ".last"][0])
        out8 = ext.dequant_int8_to_fp16(q8.contiguous(), s8)                      # ops.py:85-87
        assert out8.dtype == torch.float16 and out8.shape == q8.shape
        assert np.array_equal(bits(out8), bits(g1[fam + ".dq8.f16"])), fam
        out4 = ext.dequant_int4_packed_to_fp16(p4.contiguous(), s4, last)        # ops.py:114-117
        assert out4.shape[-1] == 2 * p4.shape[-1]
        assert np.array_equal(bits(out4[..., :last]), bits(g1[fam + ".dq4.f16"])), fam  # the call site's slice, ops.py:118-119
        n += 1
    assert n >= 24
    with pytest.raises(RuntimeError):  # the library's error string surfaces as the RuntimeError TORCH_CHECK would raise
        ext._check(ext._L.kvq_dequant_i8_f16_flat(None, 1.0, None, -1, None))
    with pytest.raises(AssertionError):
        ext.dequant_int8_to_fp16(torch.zeros(4, dtype=torch.int8), 1.0)  # CPU tensor (extensions.py:33)


def _level2_namespace(g5, t0_tokens):
    from efficient_llm_inference_amd import _lib
    kv = to_torch(g5["tiny.f16.kv"])  # [L,2,B,H,T,D]
    L, _, B, H, T, D = kv.shape
    Tcap = T + 3
    store = torch.zeros(L, B, H, Tcap, D, dtype=torch.int8, device="cuda")
    scales = torch.zeros(L, Tcap, dtype=torch.float32, device="cuda")
    ns = {"ctypes": ctypes, "torch": torch, "lib": _lib.load(), "KvqDims": _lib.KvqDims, "KvqStrides": _lib.KvqStrides, "byref": ctypes.byref,
          "KVQ_F16": _lib.dtype_code(torch.float16), "store": store, "scales": scales, "Tcap": Tcap,
          "ws": torch.empty(L * T, dtype=torch.float32, device="cuda"), "out": torch.full((L, B, H, T, D), float("nan"), dtype=torch.float16, device="cuda"),
          "stream": ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)}
    return ns, kv, (L, B, H, T, D)


def test_level2_call_sequence_against_the_reference_fixtures(g5):
    src = _blocks("Level 2")[0]
    assert "kvq_quant_i8_tokens" in src and "kvq_dequant_i8_tokens" in src
    ns, kv, (L, B, H, T, D) = _level2_namespace(g5, 0)
    # the reference's own order: the prompt's first T - 1 tokens at once (init_from_prompt_past), then one more (append_from_past)
    for t0, n in ((0, T - 1), (T - 1, 1)):
        ns.update(t0=t0, past_key_values=tuple((kv[l, 0, :, :, t0:t0 + n], kv[l, 1, :, :, t0:t0 + n]) for l in range(L)),
                  out=torch.full((L, B, H, t0 + n, D), float("nan"), dtype=torch.float16, device="cuda"))
        exec(compile(src, "INTEGRATION.md[Level 2]", "exec"), ns)
    torch.cuda.synchronize()
    assert np.array_equal(to_numpy(ns["store"][:, :, :, :T]), g5["tiny.f16.int8.kq"])           # the reference's stored int8 bytes
    assert np.array_equal(bits(ns["scales"][:, :T].half()), bits(g5["tiny.f16.int8.scales"][:, 0]))  # its stored (fp16) scales
    assert np.array_equal(bits(ns["out"]), bits(g5["tiny.f16.int8.deq"][:, 0]))                # its to_past_key_values() K


def test_level2_sharded_sequence_on_one_rank(g5):
    import torch.distributed as dist
    src1, src2 = _blocks("Level 2")[:2]
    assert "kvq_absmax_tokens" in src2 and "all_reduce" in src2
    ns, kv, (L, B, H, T, D) = _level2_namespace(g5, 0)
    ns.update(t0=0, past_key_values=tuple((kv[l, 0], kv[l, 1]) for l in range(L)))
    exec(compile(src1, "INTEGRATION.md[Level 2]", "exec"), ns)  # defines ptrs, in_st, store_st ... and the single-pass result
    torch.cuda.synchronize()
    want_store, want_scales = ns["store"].clone(), ns["scales"].clone()
    ns["store"].zero_()
    ns["scales"].zero_()
    ns["amax"] = torch.empty(L, T, dtype=torch.float32, device="cuda")
    created = False
    if not dist.is_initialized():
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        # a 1-rank group only has to make torch.distributed.all_reduce callable on the GPU table: gloo comes up in
        # milliseconds and carries CUDA tensors through the host (RCCL's own 1-rank bring-up is tests/test_sharded_batch.py's, 4 s)
        try:
            dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
            probe = torch.zeros(1, device="cuda")
            dist.all_reduce(probe, op=dist.ReduceOp.MAX)
            created = True
        except Exception:  # noqa: BLE001
            if dist.is_initialized():
                dist.destroy_process_group()
            try:
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port + 1}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
                created = True
            except Exception as exc:  # noqa: BLE001
                pytest.skip(f"no 1-rank process group here: {exc}")
    try:
        exec(compile(src2, "INTEGRATION.md[Level 2, sharded]", "exec"), ns)
        torch.cuda.synchronize()
    finally:
        if created:
            dist.destroy_process_group()
    assert torch.equal(ns["store"], want_store) and torch.equal(ns["scales"], want_scales)  # one rank: the three steps = the single pass
