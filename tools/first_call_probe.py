#!/usr/bin/env python3
"""First east_hip_build_texts call on fresh handles (arena / ring allocation included) next to the second call:
64 MiB of ASCII as one text, and 64 prose-like texts of 1 MiB (EAST_HIP_TRACE=1: where the first call's extra goes)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd")); sys.path.insert(0, ROOT)
from east import hip_backend, synthetic
hip_backend.unicode_tables()
cases = {"ascii 1 x 64 MiB": [synthetic.word_stream_document(np.random.default_rng(20242), 64 << 20)[0]],
         "prose-like 64 x 1 MiB": synthetic.prose_like_texts(np.random.default_rng(3), 64, 1 << 20)}
for name, texts in cases.items():
    for rep in range(3):
        t0 = time.perf_counter(); index = hip_backend.HipIndex(0); t1 = time.perf_counter()
        index.build_texts(texts); t2 = time.perf_counter()
        index.build_texts(texts); t3 = time.perf_counter()
        print("%s rep %d: handle %.2f ms, first call %.2f ms, second %.2f ms (prep %.2f build %.2f)"
              % (name, rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, index.last_prep_ms, index.last_build_ms), flush=True)
        index._lib.east_hip_destroy(index._h); index._h = None       # (no pooling: the next repetition starts from nothing)
