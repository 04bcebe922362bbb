#!/usr/bin/env python3
"""Randomised differential test of the device build against the CPU oracle, beyond what tests/ runs
every time: random alphabets (2..1200 symbols), one or several documents, short strings and long
strings, planted repeats (copied passages, runs of one symbol, whole strings repeated many times),
and a random choice of the code paths a test knob selects (window sort / DC3 only / 64-bit window
keys / lean).  A third of the collections go to the device with the upper part of their alphabet
lifted, order-preserving, to random code points at or above U+0A00 (tagged symbol encoding) while
the oracle sees the original.  Every table of every document bit-exact, a few scores bit-equal.

    python tools/fuzz_gpu.py [--seconds 300] [--seed 1] [--max-symbols 400000]
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


def random_string(rng, alphabet, length, repeats):
    s = rng.choice(alphabet, size=length).astype(np.uint32)
    for _ in range(repeats):
        kind = rng.integers(0, 3)
        if length < 8:
            break
        if kind == 0:                                   # a passage copied elsewhere
            k = int(rng.integers(2, max(3, length // 3)))
            a, b = int(rng.integers(0, length - k)), int(rng.integers(0, length - k))
            s[b:b + k] = s[a:a + k].copy()
        elif kind == 1:                                 # a run of one symbol
            k = int(rng.integers(2, max(3, length // 4)))
            a = int(rng.integers(0, length - k))
            s[a:a + k] = s[a]
        else:                                           # a short period
            k = int(rng.integers(2, max(3, length // 4)))
            per = int(rng.integers(1, 5))
            a = int(rng.integers(0, length - k))
            s[a:a + k] = np.resize(s[a:a + per], k)
    return s


def random_collection(rng, max_symbols):
    sigma = int(rng.choice([2, 3, 4, 8, 26, 27, 60, 120, 254, 255, 400, 1200]))
    # (code points >= 2: the reference -- and with it the oracle -- pads with chr(1) and is undefined below that)
    alphabet = (np.arange(sigma) + int(rng.choice([2, 33, 65, 0x100, 0x400]))).astype(np.uint32)
    alphabet = alphabet[alphabet < TERM]
    n_docs = int(rng.choice([1, 1, 2, 3, 7, 40]))
    budget = int(rng.integers(50, max_symbols)) // n_docs
    docs = []
    for _ in range(n_docs):
        style = rng.integers(0, 3)
        strings = []
        left = max(4, budget)
        while left > 0:
            if style == 0:
                ln = int(rng.integers(1, 40))                       # text-like short strings
            elif style == 1:
                ln = int(rng.integers(1, max(2, left)))             # few long strings
            else:
                ln = int(rng.choice([1, 5, 300, 5000, max(2, left)]))
            ln = min(ln, left)
            strings.append(random_string(rng, alphabet, ln, int(rng.integers(0, 4))))
            left -= ln + 1
            if len(strings) > 1 and rng.random() < 0.15:            # a whole string again
                strings.append(strings[int(rng.integers(0, len(strings)))].copy())
                left -= strings[-1].size + 1
        docs.append(strings)
    return docs


def to_symbols(strings):
    parts = []
    for i, s in enumerate(strings):
        parts.append(s)
        parts.append(np.array([TERM + i], dtype=np.uint32))
    return np.concatenate(parts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-symbols", type=int, default=400000)
    args = ap.parse_args()
    easa_oracle.build()
    lib = hip_backend.load()
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    cases = docs_checked = symbols = 0
    paths = {}
    while time.time() < t_end:
        docs = random_collection(rng, args.max_symbols)
        # (7 / 9: first-level keys of variable-length code words, with / without the fused finish)
        knob = int(rng.choice([1, 1, 1, 0, 3, 2, 4, 5, 7, 7, 9, 9]))
        lib.east_hip_debug_set_window_sort(knob)
        lib.east_hip_debug_set_lds_rounds(int(rng.choice([1, 1, 3, 2, 0])))
        # (the persistent rounds: the form by size, or the large form forced, on a grid of what the device holds / 3 / 40 workgroups)
        lib.east_hip_debug_set_persist(int(rng.choice([0, 0, 1])), int(rng.choice([0, 0, 3, 40])))
        lib.east_hip_debug_set_segmented_sort(int(rng.choice([-1, 1, 1, 0])))     # (1: wherever a shard holds 2 .. 65535 documents)
        lib.east_hip_debug_set_score_path(int(rng.choice([1, 4, 4, 0, 2, 3, 5])))
        parts = [to_symbols(sc) for sc in docs]
        sym = np.concatenate(parts)
        off = np.concatenate([[0], np.cumsum([p.size for p in parts])])
        # what the device is given: the same, or the upper part of the alphabet lifted above U+0A00 (tagged terminators)
        used = np.unique(sym[sym < TERM])
        lifted = rng.random() < 0.35
        if lifted:
            split = int(rng.integers(0, used.size + 1))
            target = used.copy()
            target[split:] = np.sort(rng.choice(0x110000 - TERM, size=used.size - split, replace=False).astype(np.uint32) + TERM)
            lift = lambda a: target[np.searchsorted(used, a)]      # noqa: E731
            dev_sym = np.where(sym >= TERM, (sym - TERM) | np.uint32(0x80000000), lift(np.minimum(sym, used[-1]))).astype(np.uint32)
        else:
            lift = lambda a: a                                     # noqa: E731
            dev_sym = sym
        index = hip_backend.HipIndex()
        if os.environ.get("FUZZ_VERBOSE"):
            print("case %d: knob %d, %d docs, %d symbols, strings %s" % (cases, knob, len(docs), sym.size,
                                                                         [len(sc) for sc in docs]), flush=True)
        try:
            index.build(dev_sym, off, np.array([len(sc) for sc in docs], dtype=np.int32))
            info = index.info()
            key = (knob, info["window_sorted"], min(info["dc3_levels"], 3), min(info["refine_rounds"], 3), int(lifted),
                   info["fused_finish"], info["ht_keys"], info["seg_sort"], int(info["persist_rounds"] > 0))
            paths[key] = paths.get(key, 0) + 1
            queries = []
            for sc in docs[:2]:
                s = sc[int(rng.integers(0, len(sc)))]
                if s.size:
                    a = int(rng.integers(0, s.size))
                    queries.append(s[a:a + int(rng.integers(1, 12))])
            queries.append(rng.choice(sym[sym < TERM], size=3).astype(np.uint32))
            qo = np.concatenate([[0], np.cumsum([q.size for q in queries])]).astype(np.int64)
            qs = lift(np.concatenate(queries)).astype(np.uint32)
            table = {norm: index.score_table(qs, qo, norm) for norm in (True, False)}
            for d, p in enumerate(parts):
                if p.size > 250000 and d > 0:
                    continue                                        # (the oracle is the slow side)
                o = easa_oracle.OracleEASA(symbols=p, n_strings=len(docs[d]))
                t = index.tables(d)
                for name in TABLES:
                    if not np.array_equal(t[name], getattr(o, name)):
                        np.save("/tmp/fuzz_fail_symbols.npy", sym)
                        np.save("/tmp/fuzz_fail_offsets.npy", off)
                        raise SystemExit("MISMATCH %s doc %d, knob %d, info %r (inputs saved under /tmp)" % (name, d, knob, info))
                for norm in (True, False):
                    for k, q in enumerate(queries):
                        want = o.score_symbols(q, norm, fast=True)
                        if table[norm][k, d] != want:
                            raise SystemExit("SCORE MISMATCH doc %d query %d: %r != %r, knob %d" % (d, k, table[norm][k, d], want, knob))
                docs_checked += 1
                symbols += p.size
        finally:
            index.close()
        cases += 1
    lib.east_hip_debug_set_window_sort(1)
    lib.east_hip_debug_set_lds_rounds(1)
    lib.east_hip_debug_set_persist(0, 0)
    lib.east_hip_debug_set_segmented_sort(-1)
    lib.east_hip_debug_set_score_path(1)
    print("fuzz ok: %d collections, %d documents, %d symbols checked; paths (knob, window_sorted, dc3_levels, rounds, lifted, fused, "
          "variable-length keys, segmented sort, persistent rounds):"
          % (cases, docs_checked, symbols))
    for k in sorted(paths):
        print("   ", k, paths[k])


if __name__ == "__main__":
    main()
