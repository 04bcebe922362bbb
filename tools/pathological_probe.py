#!/usr/bin/env python3
"""Build times of repetitive inputs next to random text of the same size: a run of one symbol, a short period, 16 copies of
a 1 MiB passage in one string, a Fibonacci string, the reference's worst-case collection -- is there a cliff anywhere?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ast-text-analysis_amd"))
import numpy as np
from east import hip_backend, synthetic
T = 0x0A00
n = int(sys.argv[1]) if len(sys.argv) > 1 else (16 << 20)
rng = np.random.default_rng(5)
def fib(n):
    a, b = np.array([65], np.uint32), np.array([65, 66], np.uint32)
    while b.size < n:
        a, b = b, np.concatenate([b, a])
    return b[:n]
cases = {
    "random A-Z (direct)": rng.integers(65, 91, size=n, dtype=np.uint32),
    "one symbol": np.full(n, 65, np.uint32),
    "period 3": np.resize(np.array([65, 66, 67], np.uint32), n),
    "16 copies of a passage": np.tile(rng.integers(65, 91, size=n // 16, dtype=np.uint32), 16),
    "fibonacci": fib(n),
    "2-letter random": rng.integers(65, 67, size=n, dtype=np.uint32),
}
index = hip_backend.HipIndex(0, reserve_symbols=n + 1)
for name, body in cases.items():
    sym = np.concatenate([body, [T]]).astype(np.uint32)
    times = []
    for _ in range(3):
        index.build(sym, np.array([0, sym.size]), np.array([1]))
        times.append(index.last_build_ms)
    info = index.info()
    print("%-26s n=%d: build %8.2f ms (%.3f ns/symbol)  rounds %d dc3_levels %d window_sorted %d" %
          (name, sym.size, min(times), min(times) * 1e6 / sym.size, info["refine_rounds"], info["dc3_levels"], info["window_sorted"]), flush=True)
    if os.environ.get("EAST_PROFILE"):
        index.profile_enable(True)
        index.build(sym, np.array([0, sym.size]), np.array([1]))
        prof = index.profile_report()
        index.profile_enable(False)
        for k, (c, t) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:6]:
            print("      %-46s %4d launches %9.3f ms" % (k[:46], c, t))
