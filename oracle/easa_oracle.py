"""TEST INFRASTRUCTURE ONLY -- Python face of the CPU oracle (easa_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product package never does.

`OracleEASA(strings)` mirrors the attributes and `score()` of the reference's
EnhancedAnnotatedSuffixArray (east/asts/easa.py:12-36) on top of the C
restatement.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("EASA_ORACLE_LIBRARY", os.path.join(_HERE, "libeasa_oracle.so"))   # (the variable: the sanitizer build)
TERMINATOR_START = 0x0A00  # east/consts.py:23-24

_lib = None


def build(force=False):
    """Compile libeasa_oracle.so with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "easa_oracle.c")
    if "EASA_ORACLE_LIBRARY" in os.environ:
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libeasa_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, u32p, i64p, dblp, intp = (ctypes.c_int64, ctypes.POINTER(ctypes.c_uint32),
                                       ctypes.POINTER(ctypes.c_int64),
                                       ctypes.POINTER(ctypes.c_double),
                                       ctypes.POINTER(ctypes.c_int))
        L.easa_build.argtypes = [u32p, i64, i64] + [i64p] * 6
        L.easa_build.restype = ctypes.c_int
        L.easa_suftab.argtypes = [u32p, i64, i64p]
        L.easa_suftab.restype = ctypes.c_int
        L.easa_lcptab.argtypes = [u32p, i64, i64p, i64p]
        L.easa_lcptab.restype = ctypes.c_int
        L.easa_score.argtypes = [u32p, i64] + [i64p] * 6 + [u32p, i64, ctypes.c_int, dblp, intp]
        L.easa_score.restype = ctypes.c_double
        L.easa_score_fast.argtypes = [u32p, i64, i64, i64p, u32p, i64, ctypes.c_int,
                                      dblp, i64p, intp]
        L.easa_score_fast.restype = ctypes.c_double
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def make_symbols(strings_collection):
    """make_unique_endings + "".join (east/asts/utils.py:25-40, easa.py:19) as
    a uint32 code-point array: string i is followed by U+0A00+i."""
    parts = []
    for i, s in enumerate(strings_collection):
        parts.append(np.fromiter((ord(c) for c in s), dtype=np.uint32, count=len(s)))
        parts.append(np.array([TERMINATOR_START + i], dtype=np.uint32))
    return np.concatenate(parts) if parts else np.zeros(0, np.uint32)


def query_symbols(query):
    """score() strips U+0020 only (easa.py:36)."""
    q = query.replace(" ", "")
    return np.fromiter((ord(c) for c in q), dtype=np.uint32, count=len(q))


class OracleEASA(object):
    """CPU oracle twin of EnhancedAnnotatedSuffixArray (easa.py:12-24)."""

    def __init__(self, strings_collection=None, symbols=None, n_strings=None, tables=True):
        if symbols is None:
            if not strings_collection:
                raise ValueError("empty strings collection")  # base.py:20-22
            symbols = make_symbols(strings_collection)
            n_strings = len(strings_collection)
        self.symbols = np.ascontiguousarray(symbols, dtype=np.uint32)
        self.n = int(self.symbols.shape[0])
        self.m = int(n_strings)
        n = self.n
        self.suftab = np.zeros(n, np.int64)
        self.lcptab = np.zeros(n, np.int64)
        L = lib()
        if tables:
            self.childtab_up = np.zeros(n, np.int64)
            self.childtab_down = np.zeros(n, np.int64)
            self.childtab_next_l_index = np.zeros(n, np.int64)
            self.anntab = np.zeros(n, np.int64)
            rc = L.easa_build(_p(self.symbols, ctypes.c_uint32), n, self.m,
                              _p(self.suftab, ctypes.c_int64), _p(self.lcptab, ctypes.c_int64),
                              _p(self.childtab_up, ctypes.c_int64),
                              _p(self.childtab_down, ctypes.c_int64),
                              _p(self.childtab_next_l_index, ctypes.c_int64),
                              _p(self.anntab, ctypes.c_int64))
            if rc:
                raise ValueError("easa_build failed (%d): empty input, or a symbol <= U+0001 -- the reference pads "
                                 "with chr(1), easa.py:149, and is undefined for such text" % rc)
        else:  # SA + LCP only (enough for score_fast)
            if L.easa_suftab(_p(self.symbols, ctypes.c_uint32), n, _p(self.suftab, ctypes.c_int64)):
                raise ValueError("easa_suftab failed: empty input, or a symbol <= U+0001")
            L.easa_lcptab(_p(self.symbols, ctypes.c_uint32), n,
                          _p(self.suftab, ctypes.c_int64), _p(self.lcptab, ctypes.c_int64))
            self.anntab = None

    # -- score ------------------------------------------------------------
    def score_symbols(self, q, normalized=True, fast=False, want_suffix=False, want_probes=False):
        q = np.ascontiguousarray(q, dtype=np.uint32)
        L = lib()
        err = ctypes.c_int(0)
        suf = np.zeros(max(len(q), 1), np.float64)
        probes = ctypes.c_int64(0)
        if fast or self.anntab is None:
            r = L.easa_score_fast(_p(self.symbols, ctypes.c_uint32), self.n, self.m,
                                  _p(self.suftab, ctypes.c_int64), _p(q, ctypes.c_uint32),
                                  len(q), int(bool(normalized)), _p(suf, ctypes.c_double),
                                  ctypes.byref(probes), ctypes.byref(err))
        else:
            r = L.easa_score(_p(self.symbols, ctypes.c_uint32), self.n,
                             _p(self.suftab, ctypes.c_int64), _p(self.lcptab, ctypes.c_int64),
                             _p(self.childtab_up, ctypes.c_int64),
                             _p(self.childtab_down, ctypes.c_int64),
                             _p(self.childtab_next_l_index, ctypes.c_int64),
                             _p(self.anntab, ctypes.c_int64), _p(q, ctypes.c_uint32), len(q),
                             int(bool(normalized)), _p(suf, ctypes.c_double), ctypes.byref(err))
        if err.value:
            raise ZeroDivisionError("float division by zero")  # easa.py:134 on an empty query
        out = [float(r)]
        if want_suffix:
            out.append(suf[:len(q)].copy())
        if want_probes:
            out.append(int(probes.value))
        return out[0] if len(out) == 1 else tuple(out)

    def score(self, query, normalized=True, synonimizer=None, return_suffix_scores=False,
              fast=False):
        q = query.replace(" ", "")
        qs = query_symbols(query)
        if return_suffix_scores:
            r, suf = self.score_symbols(qs, normalized, fast, want_suffix=True)
            return r, {q[i:]: float(suf[i]) for i in range(len(q))}
        return self.score_symbols(qs, normalized, fast)


# ---- closed forms, numpy, small inputs only ----------
def closed_form_tables(lcp):
    """anntab / next / up / down from lcptab via nearest-smaller values; used to
    cross-check the formulation the HIP kernels implement.  O(n * scan)."""
    lcp = np.asarray(lcp, dtype=np.int64)
    n = len(lcp)
    ann = np.zeros(n, np.int64)
    nxt = np.zeros(n, np.int64)
    up = np.zeros(n, np.int64)
    down = np.zeros(n, np.int64)
    for k in range(n):
        v = lcp[k]
        pse = k - 1
        while pse >= 0 and lcp[pse] > v:
            pse -= 1
        nse = k + 1
        while nse < n and lcp[nse] > v:
            nse += 1
        if k >= 1 and v > 0 and (pse < 0 or lcp[pse] < v):
            psv = pse
            nsv = k + 1
            while nsv < n and lcp[nsv] >= v:
                nsv += 1
            ann[k] = nsv - psv
        if nse < n and lcp[nse] == v:
            nxt[k] = nse
        if pse >= 0 and k - pse > 1:
            seg = lcp[pse + 1:k]
            up[k] = pse + 1 + int(np.argmin(seg))
        if nse < n and nse - k > 1:
            seg = lcp[k + 1:nse]
            down[k] = k + 1 + int(np.argmin(seg))
    return ann, nxt, up, down
