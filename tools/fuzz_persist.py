#!/usr/bin/env python3
"""Randomised differential test aimed at the persistent refinement rounds (csrc/persist_rounds.h): REPETITIVE
collections -- m identical or nearly identical strings (the reference's worst-case harness shape, analysis/utils.py:5-9),
passages copied inside and across strings, text over two or three letters --, one to four documents, with the form (resident
/ large) and the grid (what the device holds / a few workgroups) drawn at random (east_hip_debug_set_persist) and the
launch-by-launch rounds as a control.  Every table of every document bit-exact against the CPU oracle.

    python tools/fuzz_persist.py [--seconds 300] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
sys.path.insert(0, ROOT)
from east import hip_backend  # noqa: E402
from oracle import easa_oracle  # noqa: E402

TABLES = ("suftab", "lcptab", "anntab", "childtab_up", "childtab_down", "childtab_next_l_index")
TERM = 0x0A00


def document(rng, budget):
    """One document: (symbols with terminators, number of strings)."""
    sigma = int(rng.choice([2, 2, 3, 4, 26]))
    alphabet = (65 + np.arange(sigma)).astype(np.uint32)
    kind = int(rng.integers(0, 4))
    strings = []
    if kind == 0:                                       # m copies of one string, a few of them changed in one place
        m = int(rng.choice([2, 3, 8, 40, 100]))
        length = max(8, min(3000, budget // m))
        base = rng.choice(alphabet, size=length).astype(np.uint32)
        for _ in range(m):
            s = base.copy()
            if rng.random() < 0.2:
                s[int(rng.integers(0, length))] = rng.choice(alphabet)
            if rng.random() < 0.2:
                s = s[:int(rng.integers(1, length + 1))]
            strings.append(s)
    elif kind == 1:                                     # a passage written several times inside a few strings
        passage = rng.choice(alphabet, size=int(rng.integers(20, 1500))).astype(np.uint32)
        for _ in range(int(rng.integers(1, 6))):
            reps = int(rng.integers(1, 8))
            filler = rng.choice(alphabet, size=int(rng.integers(0, 300))).astype(np.uint32)
            strings.append(np.concatenate([np.tile(passage, reps), filler])[:max(4, budget // 4)])
    elif kind == 2:                                     # random text over a tiny alphabet: long accidental repeats, many groups
        n_str = int(rng.integers(1, 5))
        for _ in range(n_str):
            strings.append(rng.choice(alphabet[:2], size=max(8, budget // n_str)).astype(np.uint32))
    else:                                               # overlapping windows of one text
        text = rng.choice(alphabet, size=max(64, budget // 2)).astype(np.uint32)
        for _ in range(int(rng.integers(2, 5))):
            a = int(rng.integers(0, text.size // 2))
            strings.append(text[a:a + int(rng.integers(16, text.size // 2 + 17))])
    parts = []
    for s in strings:
        parts.append(s)
        parts.append(np.array([TERM + len(parts) // 2], dtype=np.uint32))
    return np.concatenate(parts), len(strings)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-symbols", type=int, default=160000)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    lib = hip_backend.load()
    easa_oracle.build()
    t_end = time.time() + args.seconds
    cases = docs_checked = symbols = 0
    forms = {}
    while time.time() < t_end:
        n_docs = int(rng.choice([1, 1, 2, 4]))
        budget = int(rng.integers(2000, args.max_symbols)) // n_docs
        docs = [document(rng, budget) for _ in range(n_docs)]
        force_large, wgs = int(rng.choice([0, 1, 1])), int(rng.choice([0, 0, 2, 7, 60]))
        rounds_knob = int(rng.choice([1, 1, 1, 3]))
        lib.east_hip_debug_set_persist(force_large, wgs)
        lib.east_hip_debug_set_lds_rounds(rounds_knob)
        lib.east_hip_debug_set_window_sort(int(rng.choice([1, 1, 4])))
        sym = np.concatenate([d[0] for d in docs])
        off = np.concatenate([[0], np.cumsum([d[0].size for d in docs])]).astype(np.int64)
        index = hip_backend.HipIndex()
        try:
            index.build(sym, off, np.array([d[1] for d in docs], dtype=np.int32))
            info = index.info()
            key = (rounds_knob, force_large, wgs, int(info["persist_rounds"] > 0))
            forms[key] = forms.get(key, 0) + 1
            for d, (p, m) in enumerate(docs):
                o = easa_oracle.OracleEASA(symbols=p, n_strings=m)
                t = index.tables(d)
                for name in TABLES:
                    if not np.array_equal(t[name], getattr(o, name)):
                        np.save("/tmp/fuzz_persist_fail.npy", p)
                        raise SystemExit("MISMATCH %s in document %d of case %d (seed %d): knobs %r, info %r"
                                         % (name, d, cases, args.seed, key, info))
                docs_checked += 1
                symbols += p.size
        finally:
            index.close()
        cases += 1
    lib.east_hip_debug_set_persist(0, 0)
    lib.east_hip_debug_set_lds_rounds(1)
    lib.east_hip_debug_set_window_sort(1)
    print("fuzz_persist ok: %d collections, %d documents, %d symbols; (rounds knob, large form forced, grid cap, persistent rounds ran): count"
          % (cases, docs_checked, symbols))
    for k in sorted(forms):
        print("   ", k, forms[k])


if __name__ == "__main__":
    main()
