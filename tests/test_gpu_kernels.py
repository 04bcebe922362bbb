"""gpu tier: the sort / scan / DC3 kernels on their own, through the C ABI's
kernel-level entry points, against numpy and the oracle."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _sort(hip, keys, vals, bits):
    lib = hip.load()
    fn = lib.east_hip_debug_radix_sort_u64 if keys.dtype == np.uint64 else lib.east_hip_debug_radix_sort_u32
    ct = ctypes.c_uint64 if keys.dtype == np.uint64 else ctypes.c_uint32
    rc = fn(0, _p(keys, ct), _p(vals, ctypes.c_uint32), keys.size, bits)
    assert rc == 0, lib.east_hip_last_error()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4095, 4096, 4097, 12345, 1 << 20, (1 << 22) + 77])
@pytest.mark.parametrize("dtype,bits", [(np.uint32, 8), (np.uint32, 15), (np.uint32, 32), (np.uint64, 45), (np.uint64, 64)])
def test_radix_sort_stable(hip, n, dtype, bits):
    rng = np.random.default_rng(n * 131 + bits)
    # few distinct values on purpose: stability is only visible with duplicates
    hi = min(1 << bits, 1 << 62)
    keys = rng.integers(0, hi, size=n, dtype=np.uint64)
    if n > 100:
        keys[rng.integers(0, n, size=n // 2)] = keys[0]
    keys = keys.astype(dtype)
    vals = np.arange(n, dtype=np.uint32)
    k2, v2 = keys.copy(), vals.copy()
    _sort(hip, k2, v2, bits)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k2, keys[order])
    assert np.array_equal(v2, vals[order])


def test_radix_sort_partial_bits_ignores_high_bits(hip):
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 1 << 40, size=100000, dtype=np.uint64)
    vals = np.arange(keys.size, dtype=np.uint32)
    k2, v2 = keys.copy(), vals.copy()
    _sort(hip, k2, v2, 16)
    order = np.argsort(keys & np.uint64(0xFFFF), kind="stable")
    assert np.array_equal(v2, vals[order])


@pytest.mark.parametrize("n", [1, 2, 255, 256, 4096, 4097, 100000, (1 << 24) + 3])
def test_exclusive_scan(hip, n):
    rng = np.random.default_rng(n)
    a = rng.integers(0, 7, size=n, dtype=np.uint32)
    out = np.empty_like(a)
    lib = hip.load()
    rc = lib.east_hip_debug_exclusive_scan(0, _p(a, ctypes.c_uint32), _p(out, ctypes.c_uint32), n)
    assert rc == 0, lib.east_hip_last_error()
    ref = np.concatenate([[0], np.cumsum(a[:-1], dtype=np.uint64)]).astype(np.uint32)
    assert np.array_equal(out, ref)


def _sa(hip, s, sigma):
    out = np.empty(s.size, dtype=np.int32)
    lv = ctypes.c_int32(0)
    lib = hip.load()
    rc = lib.east_hip_debug_suffix_array(0, _p(s, ctypes.c_uint32), s.size, sigma, _p(out, ctypes.c_int32),
                                         ctypes.byref(lv))
    assert rc == 0, lib.east_hip_last_error()
    return out, lv.value


def _naive_sa(s):
    lst = s.tolist()
    return np.array(sorted(range(len(lst)), key=lambda i: lst[i:]), dtype=np.int32)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 10, 31, 100, 1000])
@pytest.mark.parametrize("sigma", [1, 2, 3, 26])
def test_suffix_array_small_vs_naive(hip, n, sigma):
    """Plain strings without a unique last symbol: DC3 with the 0 pad must order
    'shorter prefix first' exactly like list comparison."""
    rng = np.random.default_rng(n * 7 + sigma)
    for _ in range(3):
        s = rng.integers(1, sigma + 1, size=n, dtype=np.uint32)
        sa, _ = _sa(hip, s, sigma)
        assert np.array_equal(sa, _naive_sa(s)), (s.tolist(), sa.tolist())


@pytest.mark.parametrize("case", ["all_same", "periodic2", "periodic3", "fibonacci", "wide_alphabet"])
def test_suffix_array_adversarial(hip, case):
    n = 20000
    if case == "all_same":
        s = np.ones(n, dtype=np.uint32)
    elif case == "periodic2":
        s = np.tile(np.array([1, 2], dtype=np.uint32), n // 2)
    elif case == "periodic3":
        s = np.tile(np.array([2, 1, 1], dtype=np.uint32), n // 3)
    elif case == "fibonacci":
        a, b = [1], [1, 2]
        while len(b) < n:
            a, b = b, b + a
        s = np.array(b[:n], dtype=np.uint32)
    else:   # distinct symbols beyond 2^21: the two-stage (3b > 64) key path
        rng = np.random.default_rng(3)
        s = rng.integers(1, 1 << 23, size=n, dtype=np.uint32)
    sigma = int(s.max())
    sa, levels = _sa(hip, s, sigma)
    assert np.array_equal(sa, _naive_sa(s))
    if case == "all_same":
        assert levels > 3            # deep recursion exercised


def test_suffix_array_vs_oracle_1m(hip, oracle):
    """Random 1 Mi symbols over 26 letters + unique last symbol, against the oracle's DC3."""
    rng = np.random.default_rng(11)
    n = 1 << 20
    s = rng.integers(1, 27, size=n, dtype=np.uint32)
    s[-1] = 27
    sa, levels = _sa(hip, s, 27)
    ref = np.zeros(n, dtype=np.int64)
    oracle.lib().easa_suftab(_p(np.ascontiguousarray(s + 1), ctypes.c_uint32), n, _p(ref, ctypes.c_int64))
    assert np.array_equal(sa.astype(np.int64), ref)
    assert levels >= 2
