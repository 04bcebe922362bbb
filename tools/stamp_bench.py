#!/usr/bin/env python3
"""Diagnostic (library built with -DRS_STAMP): where a downsweep tile spends its cycles.
Usage on the GPU box: EAST_HIP_LIBRARY=build/variants/lib_stamp.so python3 tools/stamp_bench.py [digit bits]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
from east import hip_backend
lib = hip_backend.load()
lib.east_hip_debug_read_stamps.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
n = 61_000_000
rng = np.random.default_rng(1)
keys = rng.integers(0, 1 << 30, size=n, dtype=np.uint32)
vals = np.arange(n, dtype=np.uint32)
buf = (ctypes.c_uint64 * 16)()
names = ["zero+barrier", "load wait", "ranking", "barrier", "prefix+barrier", "key scatter+barrier", "key out", "val scatter+out",
         "end barrier"]
for bits in (30, 32):
    k, v = keys.copy(), vals.copy()
    lib.east_hip_debug_read_stamps(buf, 1)
    rc = lib.east_hip_debug_radix_sort_u32(0, k.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                            v.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n, bits)
    assert rc == 0
    lib.east_hip_debug_read_stamps(buf, 1)
    tot = sum(buf[i] for i in range(9))
    print("bits", bits, "total cycles (thread 0 of every workgroup, all passes)", tot)
    for i, nm in enumerate(names):
        print("  %-22s %5.1f %%" % (nm, 100.0 * buf[i] / tot))
