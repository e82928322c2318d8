"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path.

``-m "not gpu"`` : oracle vs golden vectors, host logic, C-ABI loads + exports (no compute).
``-m gpu``       : parity tests proper, through the C-ABI on a real MI355X.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "benchcli: one real `python bench.py` run as a subprocess on the GPU box, checked field by field against the "
                                       "driver's contract (`pytest -m benchcli`, ~25 s; kept out of `-m gpu` so that the parity suite stays under a minute)")
    config.addinivalue_line("markers", "ab: variant-equality tests of kernels that lost a measurement; they need the A-B library "
                                       "(make -C efficient-llm-inference_amd/csrc ab) and run with `pytest -m ab` on the GPU box only")
    if _selects_ab(config) and not os.environ.get("KVQ_HIP_LIB"):
        # `pytest -m ab`: the A-B library is what gets loaded (before anything imports the package's _lib)
        os.environ["KVQ_HIP_LIB"] = os.path.join(ROOT, "efficient-llm-inference_amd", "lib", "ab", "libkvq_hip.so")


def _selects_ab(config) -> bool:
    """True when the -m expression POSITIVELY names the `ab` marker (`-m ab`, `-m "gpu and ab"`); `-m "not ab"` and
    `-m "gpu and not ab"` do not: they keep the shipped library and deselect the ab tests like `-m gpu` does."""
    words = (config.getoption("-m") or "").replace("(", " ").replace(")", " ").split()
    return any(w == "ab" and (i == 0 or words[i - 1] != "not") for i, w in enumerate(words))


def pytest_collection_modifyitems(config, items):
    """`ab` tests run only when the -m expression names them: `-m gpu` (the driver's run) and `-m "not gpu"` see the
    shipped library's tests only."""
    words = (config.getoption("-m") or "").replace("(", " ").replace(")", " ").split()
    wants_cli = any(w == "benchcli" and (i == 0 or words[i - 1] != "not") for i, w in enumerate(words))
    keep, drop = [], []
    for it in items:
        gone = (it.get_closest_marker("ab") and not _selects_ab(config)) or (it.get_closest_marker("benchcli") and not wants_cli)
        (drop if gone else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def g1():
    return load_golden("g1_slices.npz")


@pytest.fixture(scope="session")
def g2():
    return load_golden("g2_kat.npz")


@pytest.fixture(scope="session")
def g5():
    return load_golden("g5_cache.npz")


@pytest.fixture(scope="session")
def g6():
    return load_golden("g6_evict.npz")
