"""build_texts of D documents of S MiB (word stream): wall, prep, build -- with EAST_HIP_TRACE=1 the chunk timeline.
usage: texts_upload_probe.py D S_MiB"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ast-text-analysis_amd"))
import numpy as np
from east import hip_backend, synthetic
D, S = int(sys.argv[1]), float(sys.argv[2])
rng = np.random.default_rng(1)
texts = [synthetic.word_stream_document(rng, int(S * (1 << 20)))[0] for _ in range(D)]
hip_backend.unicode_tables()
index = hip_backend.HipIndex(0)
for rep in range(4):
    t0 = time.perf_counter()
    index.build_texts(texts)
    wall = (time.perf_counter() - t0) * 1e3
    print("D=%d x %g MiB rep %d: wall %.2f ms, prep %.2f, build %.2f" % (D, S, rep, wall, index.last_prep_ms, index.last_build_ms), flush=True)
