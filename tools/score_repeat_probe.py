#!/usr/bin/env python3
"""Does the score walk of an index that fits the 256 MB Infinity Cache run faster when it is walked again right away?
Builds D documents of 1 MiB once, then scores the same 10 000 keyphrases several times in a row (no build in between) and
prints the walk kernel's time per call.  python tools/score_repeat_probe.py [D ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
from east import hip_backend, synthetic  # noqa: E402


def main():
    for docs in [int(x) for x in sys.argv[1:]] or [16, 256]:
        rng = np.random.default_rng(7)
        parts, ms = [], []
        for _ in range(docs):
            _, sym, m = synthetic.word_stream_document(rng, 1 << 20, want_text=False)
            parts.append(sym)
            ms.append(m)
        symbols = np.concatenate(parts)
        off = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.int64)
        qs, qo = synthetic.keyphrases(rng, symbols, 10000, off)
        index = hip_backend.HipIndex()
        index.build(symbols, off, np.array(ms, dtype=np.int32))
        index.set_keyphrases(qs, qo)
        times = []
        for rep in range(6):
            index.profile_enable(True)
            index.score_resident(True)
            index.synchronize()
            rep_ms = index.profile_report().get("score_walk_kernel", (0, 0.0))[1]
            index.profile_enable(False)
            times.append(rep_ms)
        print("docs %d: walk ms per call %s  (us/doc %s)" % (docs, [round(t, 4) for t in times], [round(1e3 * t / docs, 2) for t in times]))
        index.close()


if __name__ == "__main__":
    main()
