#!/usr/bin/env python3
"""Where a SMALL build's time goes: the reference's worst-case harness collections (analysis/utils.py:5-9; 100 identical
strings) at n = 1000 / 10000 and random text of the same size -- wall clock of east_hip_build_device, device time, the
number of launches and the per-kernel sums (every kernel bracketed in a second pass).
    python tools/small_case_probe.py [n ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
import torch  # noqa: E402
from east import hip_backend, synthetic  # noqa: E402


def run(name, sym, m):
    dev = torch.device("cuda", 0)
    off, ms = np.array([0, sym.size], dtype=np.int64), np.array([m], dtype=np.int32)
    d_sym = torch.from_numpy(sym.view(np.int32)).to(dev)
    index = hip_backend.HipIndex(0, reserve_symbols=int(sym.size))
    walls, devs = [], []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        index.build_device(d_sym.data_ptr(), sym.size, off, ms)
        walls.append((time.perf_counter() - t0) * 1e3)
        devs.append(index.last_build_ms)
    info = index.info()
    index.profile_enable(True)
    index.build_device(d_sym.data_ptr(), sym.size, off, ms)
    prof = index.profile_report()
    index.profile_enable(False)
    launches = sum(v[0] for v in prof.values())
    kernel_ms = sum(v[1] for v in prof.values())
    print("%-28s symbols %8d  wall ms first %.3f min %.3f  device ms min %.3f  launches %d  kernel ms %.3f  rounds %d lds_sorted %d"
          % (name, sym.size, walls[0], min(walls[1:]), min(devs[1:]), launches, kernel_ms, info["refine_rounds"], info.get("lds_sorted", 0)))
    for k, (c, t) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:14]:
        print("      %-40s x%-4d %.4f ms" % (k, c, t))
    index.close()


def main():
    ns = [int(a) for a in sys.argv[1:]] or [1000, 10000]
    for n in ns:
        rng = np.random.default_rng(20240 + 6)
        sym, m = synthetic.worst_case_collection(rng, 100, n)
        run("worst case n=%d" % n, sym, m)
        _, rsym, rm = synthetic.word_stream_document(np.random.default_rng(20240 + 7), int(sym.size * 1.07), want_text=False)
        run("random text, same size", rsym, rm)


if __name__ == "__main__":
    main()
