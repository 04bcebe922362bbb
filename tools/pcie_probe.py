#!/usr/bin/env python3
"""What the link between host and device moves: H2D copies of the bench document's 244 MB of symbols (4 B each) out of
pageable and out of pinned host memory, next to east_hip_build (host symbols in, index out) and east_hip_build_device
(symbols already resident).  If the pinned copy alone takes what the host-resident build has over the resident one, no
staging scheme can help: the entry point is bound by 4 bytes per symbol over PCIe."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
import numpy as np
import torch
from east import hip_backend, synthetic
_, sym, m = synthetic.word_stream_document(np.random.default_rng(20242), 64 << 20, want_text=False)
n = sym.size
host = torch.from_numpy(sym.view(np.int32))
pinned = torch.empty(n, dtype=torch.int32, pin_memory=True); pinned.copy_(host)
dev = torch.empty(n, dtype=torch.int32, device="cuda")
def timed(f, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)
t_page = timed(lambda: dev.copy_(host))
t_pin = timed(lambda: dev.copy_(pinned, non_blocking=True))
index = hip_backend.HipIndex(0, reserve_symbols=n)
off, ms = np.array([0, n], dtype=np.int64), np.array([m], dtype=np.int32)
index.build(sym, off, ms)
t_host = timed(lambda: index.build(sym, off, ms))
t_dev = timed(lambda: index.build_device(dev.data_ptr(), n, off, ms))
mb = n * 4 / 1e6
print("%d symbols = %.0f MB: H2D from pageable memory %.2f ms (%.1f GB/s), from pinned memory %.2f ms (%.1f GB/s)" % (n, mb, t_page, mb / t_page, t_pin, mb / t_pin))
print("east_hip_build (host symbols) %.2f ms wall, east_hip_build_device (resident) %.2f ms wall: the difference %.2f ms" % (t_host, t_dev, t_host - t_dev))
