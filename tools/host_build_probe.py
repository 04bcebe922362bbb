import os, sys, time
sys.path.insert(0, os.path.join(os.getcwd(), "ast-text-analysis_amd"))
import numpy as np
from east import hip_backend, synthetic
_, sym, m = synthetic.word_stream_document(np.random.default_rng(20242), 64 << 20, want_text=False)
n = sym.size
index = hip_backend.HipIndex(0, reserve_symbols=n)
off, ms = np.array([0, n], dtype=np.int64), np.array([m], dtype=np.int32)
index.build(sym, off, ms); time.sleep(0.3)
ts = []
for _ in range(24):
    t0 = time.perf_counter(); index.build(sym, off, ms); ts.append((time.perf_counter() - t0) * 1e3)
print(os.environ.get("EAST_HIP_SYMBOL_THREADS"), " ".join("%.2f" % t for t in ts), "| median %.2f" % sorted(ts)[len(ts)//2])
