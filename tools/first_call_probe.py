#!/usr/bin/env python3
"""First east_hip_build_texts call on fresh handles (arena allocation included), streamed and in one piece."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd")); sys.path.insert(0, ROOT)
from east import hip_backend, synthetic
hip_backend.unicode_tables()
text = synthetic.word_stream_document(np.random.default_rng(20242), 64 << 20)[0]
lib = hip_backend.load()
for knob in (-1, 0, -1, 0):
    lib.east_hip_debug_set_text_stream(knob)
    for rep in range(3):
        index = hip_backend.HipIndex(0)
        t0 = time.perf_counter(); index.build_texts([text]); t1 = time.perf_counter()
        index.build_texts([text]); t2 = time.perf_counter()
        print("stream %d rep %d: first call %.2f ms, second %.2f ms (prep %.2f build %.2f)" % (knob, rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, index.last_prep_ms, index.last_build_ms))
        index.close()
