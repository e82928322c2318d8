"""Shared helpers for the parity tests (numpy <-> torch, dtype codes, seeded inputs)."""
import numpy as np
import torch

TD = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
NAME = {v: k for k, v in TD.items()}


def to_torch(a: np.ndarray, dtype: str = None, device="cuda") -> torch.Tensor:
    """numpy -> torch on `device`; bf16 travels as uint16 bit patterns."""
    if dtype == "bf16":
        return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).to(device)
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def to_numpy(t: torch.Tensor) -> np.ndarray:
    t = t.detach().contiguous().cpu()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    return t.numpy().copy()


def bits(a) -> np.ndarray:
    """byte view for bit-exact comparison (distinguishes -0.0 from +0.0)"""
    if isinstance(a, torch.Tensor):
        a = to_numpy(a)
    return np.ascontiguousarray(a).view(np.uint8)


def odt(dtype: str):
    """oracle dtype argument"""
    return "bf16" if dtype == "bf16" else None


def seeded_kv(shape, dtype: str, seed: int, dist: str = "normal") -> np.ndarray:
    """Synthetic KV as numpy in the oracle's representation (bf16 -> uint16 bits)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(shape, dtype=np.float32)
    if dist == "heavy":  # 1 % of values x10: outlier channels typical of K
        x = np.where(rng.random(shape) < 0.01, x * 10.0, x).astype(np.float32)
    elif dist == "tiny":  # drives fp16 stored scales to 0 and results to +-0
        x = (x * 1e-6).astype(np.float32)
    if dtype == "f32":
        return x
    if dtype == "f16":
        return x.astype(np.float16)
    from oracle import kvq_oracle as O
    return O.f32_to_bf16_bits(x)


class tunables:
    """``with tunables(quant_block=256, ...):`` — set library knobs for the block and restore what the library had
    (not 0) afterwards. A-B keys raise KvqError in the default library: tests that use them carry ``@pytest.mark.ab``
    and run against lib/ab/libkvq_hip.so (``pytest -m ab``)."""

    def __init__(self, **kv):
        self.kv = kv
        self.old = {}

    def __enter__(self):
        from efficient_llm_inference_amd import _lib
        try:
            for k, v in self.kv.items():
                old = _lib.get_tunable(k)
                _lib.set_tunable(k, int(v))
                self.old[k] = old
        except Exception:
            self.__exit__(None, None, None)
            raise
        return self

    def __exit__(self, *exc):
        from efficient_llm_inference_amd import _lib
        for k, v in self.old.items():
            _lib.set_tunable(k, v)
        self.old = {}
        return False


class CallRecorder:
    """The model as a benchmarker calls it, noting (tokens fed, cache length seen) per forward — how
    tests/golden/make_golden.py::gen_benchmarker recorded the REFERENCE's loops (g9_benchmarker.npz ``*.calls``)."""

    def __init__(self, model):
        self.__dict__["m"] = model
        self.__dict__["calls"] = []
        self.__dict__["fed"] = []  # part C of the fixture: last token fed and fp32 next-token logits of every forward
        self.__dict__["logits"] = []

    def __getattr__(self, name):
        return getattr(self.__dict__["m"], name)

    def __call__(self, *a, **kw):
        past = kw.get("past_key_values")
        plen = 0
        if past is not None:
            plen = int(past.get_seq_length()) if hasattr(past, "get_seq_length") else int(past[0][0].size(2))
        self.calls.append((int(kw["input_ids"].shape[-1]), plen))
        res = self.__dict__["m"](*a, **kw)
        self.fed.append(int(kw["input_ids"][0, -1]))
        self.logits.append(res.logits[0, -1, :].detach().to(torch.float32).cpu().numpy().copy())
        return res

    def clear(self):
        self.calls.clear()
        self.fed.clear()
        self.logits.clear()


def value_kinds(d) -> str:
    """one letter per dict value: n None, a NaN, f float, i int, s str (the fixture's ``*.kinds``)"""
    import math
    out = []
    for v in d.values():
        out.append("n" if v is None else "s" if isinstance(v, str) else "i" if isinstance(v, (int, np.integer)) and not isinstance(v, bool)
                   else "a" if isinstance(v, float) and math.isnan(v) else "f")
    return "".join(out)
