#!/usr/bin/env python3
"""Build time for text over a WIDE alphabet (more than 254 distinct text symbols: dense u32 codes, DC3), e.g. CJK:
the word stream of bench.py with its 26 letters spread, order-preserving, over `--sigma` code points from U+4E00 on
(tagged symbol encoding), next to the same text over 26 letters (byte stream, window sort).

    python tools/wide_alphabet_bench.py [--mib 16] [--sigma 3000]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
from east import hip_backend, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mib", type=float, default=16.0)
ap.add_argument("--sigma", type=int, default=3000)
args = ap.parse_args()
rng = np.random.default_rng(5)
_, sym, m = synthetic.word_stream_document(rng, int(args.mib * (1 << 20)), want_text=False)
n = sym.size
# every letter becomes one of sigma/26 code points of "its" stretch (order-preserving between letters)
per = max(args.sigma // 26, 1)
text = sym < 0x0A00
letters = np.where(text, sym - 65, 0).astype(np.int64)
wide = np.where(text, 0x4E00 + letters * per + rng.integers(0, per, size=n), (sym - 0x0A00) | 0x80000000).astype(np.uint32)
off = np.array([0, n], dtype=np.int64)
for name, s in (("26 letters (byte stream, window sort)", sym), ("%d code points from U+4E00 (u32 codes, DC3)" % (per * 26), wide)):
    index = hip_backend.HipIndex()
    index.build(s, off, np.array([m]))
    times = []
    for _ in range(3):
        index.build(s, off, np.array([m]))
        times.append(index.last_build_ms)
    info = index.info()
    print("%-52s %8d symbols  sigma %5d  build %7.2f ms  %.2e symbols/s  (window_sorted %d, dc3 levels %d)"
          % (name, n, info["sigma_text"], min(times), n / (min(times) * 1e-3), info["window_sorted"], info["dc3_levels"]))
    index.close()
    if os.environ.get("EAST_PROFILE") and name.startswith("%d code" % (per * 26)):
        index = hip_backend.HipIndex()
        index.build(s, off, np.array([m]))
        index.profile_enable(True)
        index.build(s, off, np.array([m]))
        report = index.profile_report()
        for kname, (count, ms) in sorted(report.items(), key=lambda kv: -kv[1][1])[:22]:
            print("    %-44s %3d launches %8.3f ms" % (kname, count, ms))
        index.close()
