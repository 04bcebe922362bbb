#!/usr/bin/env python3
"""Per-kernel times of a build that waits for the device's answers (a handle's first build: alphabet and planning
sample read back, placement counts read back) next to the speculative steady state, on bench.py's default document."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
import torch  # noqa: E402
from east import hip_backend, synthetic  # noqa: E402

mib = float(sys.argv[1]) if len(sys.argv) > 1 else 64.0
_, sym, m = synthetic.word_stream_document(np.random.default_rng(20240 + 2), int(mib * (1 << 20)), want_text=False)
off, ms = np.array([0, sym.size], dtype=np.int64), np.array([m], dtype=np.int32)
d_sym = torch.from_numpy(sym.view(np.int32)).to("cuda:0")
lib = hip_backend.load()
for label, spec in (("first build (no guesses)", 0), ("steady state (speculative)", 1)):
    lib.east_hip_debug_set_speculation(spec)
    index = hip_backend.HipIndex(0, reserve_symbols=int(sym.size))
    for _ in range(3):
        index.build_device(d_sym.data_ptr(), int(sym.size), off, ms)
    plain = index.last_build_ms
    index.profile_enable(True)
    for _ in range(3):
        index.build_device(d_sym.data_ptr(), int(sym.size), off, ms)
    rep = index.profile_report()
    index.profile_enable(False)
    total = sum(v[1] for v in rep.values()) / 3
    print("%s: build %.3f ms (events around every kernel: %.3f ms of kernels per build)" % (label, plain, total))
    for name, (count, t) in sorted(rep.items(), key=lambda kv: -kv[1][1])[:16]:
        print("    %-34s %3d x %7.4f ms" % (name, count // 3, t / count))
    index.close()
lib.east_hip_debug_set_speculation(1)
