#!/usr/bin/env python3
"""Per-kernel breakdown of a build of 16 Mi symbols with long repeats in ONE string (get_ast([one string])): a run of one
letter, a period of three, a Fibonacci string -- the inputs whose tie groups are longer than a tile of the persistent rounds
takes, so that the refinement goes launch by launch through prefix doubling.
    python tools/long_repeat_profile.py [one_letter|period3|fibonacci ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
import torch  # noqa: E402
from east import hip_backend  # noqa: E402

n = 16 << 20
fa, fb = np.array([65], np.uint32), np.array([65, 66], np.uint32)
while fb.size < n:
    fa, fb = fb, np.concatenate([fb, fa])
shapes = {"one_letter": np.full(n, 65, np.uint32), "period3": np.resize(np.array([65, 66, 67], np.uint32), n), "fibonacci": fb[:n]}
index = hip_backend.HipIndex(0, reserve_symbols=n + 1)
for name in (sys.argv[1:] or list(shapes)):
    sym = np.concatenate([shapes[name], [0x0A00]]).astype(np.uint32)
    d = torch.from_numpy(sym.view(np.int32)).to("cuda:0")
    off, ms = np.array([0, sym.size]), np.array([1])
    t = []
    for _ in range(3):
        index.build_device(d.data_ptr(), sym.size, off, ms)
        t.append(index.last_build_ms)
    info = index.info()
    index.profile_enable(True)
    index.build_device(d.data_ptr(), sym.size, off, ms)
    prof = index.profile_report()
    index.profile_enable(False)
    print("%s: %.1f ms, %d rounds (%d persistent), %d launches, kernel ms %.1f" % (name, min(t), info["refine_rounds"], info["persist_rounds"],
                                                                                sum(v[0] for v in prof.values()), sum(v[1] for v in prof.values())))
    for k, (c, tms) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:16]:
        print("      %-44s x%-4d %8.3f ms" % (k, c, tms))
