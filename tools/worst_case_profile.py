#!/usr/bin/env python3
"""Per-kernel times of one build of the reference's worst-case harness input (analysis/utils.py:5-9: m identical strings
of n - 4 letters) -- which path runs and where its time goes.  usage: worst_case_profile.py [n [m [docs]]]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
from east import hip_backend, synthetic      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n_docs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(20240 + 6)
docs = [synthetic.worst_case_collection(rng, m, n) for _ in range(n_docs)]
sym = np.concatenate([d[0] for d in docs])
off = np.concatenate([[0], np.cumsum([d[0].size for d in docs])]).astype(np.int64)
ms = np.array([d[1] for d in docs], dtype=np.int32)
if os.environ.get("EAST_WC_KNOB"):                       # e.g. 0 = DC3 only (east_hip_debug_set_window_sort)
    hip_backend.load().east_hip_debug_set_window_sort(int(os.environ["EAST_WC_KNOB"]))
index = hip_backend.HipIndex(0, reserve_symbols=int(sym.size))
index.build(sym, off, ms)
t0 = time.perf_counter()
index.build(sym, off, ms)
wall = (time.perf_counter() - t0) * 1e3
print("n=%d m=%d docs=%d symbols=%d: build %.2f ms (device), %.2f wall incl. H2D" % (n, m, n_docs, sym.size, index.last_build_ms, wall))
info = index.info()
print({k: info[k] for k in ("window_sorted", "refine_rounds", "dc3_levels", "radix_passes", "lds_sorted", "fused_finish", "first_kept", "first_n")})
index.profile_enable(True)
index.build(sym, off, ms)
prof = index.profile_report()
index.profile_enable(False)
total = sum(v[1] for v in prof.values())
print("kernel time %.2f ms in %d launches" % (total, sum(v[0] for v in prof.values())))
for k, (c, t) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-50s %5d launches %9.3f ms %5.1f%%" % (k[:50], c, t, 100 * t / total))
