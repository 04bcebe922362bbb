"""Per-kernel times of build_texts on the bench's prose case (64 documents of 1 MiB drawn from the image's prose)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ast-text-analysis_amd"))
import numpy as np
from east import hip_backend, synthetic
raw, _ = synthetic.image_prose(24 << 20, False)
rng = np.random.default_rng(20245)
lines = [ln for ln in raw.split(b"\n") if len(ln) > 20]
lens = np.array([len(ln) + 1 for ln in lines])
picks = rng.integers(0, len(lines), size=int((64 << 20) / lens.mean()) + 1)
big = b"\n".join(lines[i] for i in picks)[:64 << 20]
texts = [big[i:i + (1 << 20)] for i in range(0, len(big), 1 << 20)]
hip_backend.unicode_tables()
index = hip_backend.HipIndex(0)
for _ in range(3):
    t0 = time.perf_counter(); index.build_texts(texts); wall = (time.perf_counter() - t0) * 1e3
print("wall %.2f prep %.2f build %.2f" % (wall, index.last_prep_ms, index.last_build_ms), index.info())
index.profile_enable(True)
index.build_texts(texts)
prof = index.profile_report()
index.profile_enable(False)
tot = sum(v[1] for v in prof.values())
print("kernel time %.2f ms, %d launches" % (tot, sum(v[0] for v in prof.values())))
for k, (c, t) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-52s %4d %8.3f ms %5.1f%%" % (k[:52], c, t, 100 * t / tot))
