"""Test configuration.

Tiers:
  -m "not gpu"  oracle vs golden fixtures, host logic, C-ABI exports, gloo world_size-2
  -m gpu        parity tests proper: HIP path (through the C ABI) vs oracle / fixtures
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ast-text-analysis_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import easa_oracle
    easa_oracle.build()
    return easa_oracle


@pytest.fixture(scope="session")
def hip():
    """The ctypes binding; GPU tests fail (not skip) if the library or device is missing."""
    try:
        import torch  # noqa: F401  (first: torch brings its own HIP runtime and finds no GPU if the system's was loaded before it)
    except ImportError:
        pass
    from east import hip_backend
    if not os.path.exists(hip_backend.LIB_PATH):      # a fresh checkout: compile the library first
        import __graft_entry__
        __graft_entry__.build()
    hip_backend.load()
    assert hip_backend.device_count() >= 1, "no HIP device: the gpu tier must run on the GPU box"
    return hip_backend


def word_stream(rng, n_bytes, lo=3, hi=10):
    """BASELINE synthetic text: uniform A-Z words of length U{lo..hi}, single spaces."""
    n_words = n_bytes // ((lo + hi) // 2) + 16
    lens = rng.integers(lo, hi + 1, size=n_words)
    total = int(lens.sum() + n_words)
    buf = rng.integers(65, 91, size=total, dtype=np.uint8)
    ends = np.cumsum(lens + 1) - 1
    buf[ends] = 32
    return buf[:n_bytes].tobytes()
