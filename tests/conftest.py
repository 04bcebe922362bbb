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


class Renamed(object):
    """A strings collection with its text alphabet renamed, order-preserving, into code points 2, 3, ... below the
    terminator base U+0A00 -- the form in which easa.py (and the oracle, its restatement) is defined for text of any
    script: the method does not care what the symbols are called, the reference only needs its terminators
    chr(0x0A00+i) to sort above the text.  `alphabet`: every code point that must get a name (default: those of
    the strings)."""

    def __init__(self, strings, alphabet=None):
        cps = sorted(set(ord(c) for s in strings for c in s) if alphabet is None else set(alphabet))
        assert len(cps) + 3 < 0x0A00
        self.name = {c: 2 + i for i, c in enumerate(cps)}
        self.unused = 2 + len(cps)                     # what a query symbol absent from the text becomes
        parts = []
        for i, s in enumerate(strings):
            parts.append(np.array([self.name[ord(c)] for c in s] + [0x0A00 + i], dtype=np.uint32))
        self.symbols = np.concatenate(parts)
        self.n_strings = len(strings)

    def query(self, q):
        """score() strips U+0020 only (easa.py:36) -- before the renaming."""
        return np.array([self.name.get(ord(c), self.unused) for c in q.replace(" ", "")], dtype=np.uint32)


def word_stream(rng, n_bytes, lo=3, hi=10):
    """BASELINE synthetic text: uniform A-Z words of length U{lo..hi}, single spaces."""
    n_words = n_bytes // ((lo + hi) // 2) + 16
    lens = rng.integers(lo, hi + 1, size=n_words)
    total = int(lens.sum() + n_words)
    buf = rng.integers(65, 91, size=total, dtype=np.uint8)
    ends = np.cumsum(lens + 1) - 1
    buf[ends] = 32
    return buf[:n_bytes].tobytes()
