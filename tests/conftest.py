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
