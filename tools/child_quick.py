"""On the GPU box: the two child-table kernels on the 64 MiB bench document and on 256 x 1 MiB documents (per-kernel times)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "ast-text-analysis_amd"))
import numpy as np
from east import hip_backend, synthetic
for docs, mib in ((1, 64), (256, 1)):
    rng = np.random.default_rng(5)
    parts = [synthetic.word_stream_document(rng, mib << 20, want_text=False)[1:] for _ in range(docs)]
    sym = np.concatenate([p[0] for p in parts])
    off = np.concatenate([[0], np.cumsum([p[0].size for p in parts])])
    index = hip_backend.HipIndex()
    index.build(sym, off, np.array([p[1] for p in parts]))
    index.profile_enable(True)
    index.tables(0, names=("childtab_up",))
    rep = index.profile_report()
    index.profile_enable(False)
    print(docs, "x", mib, "MiB:", {k: round(v[1] / v[0], 3) for k, v in rep.items() if k.startswith("child")})
