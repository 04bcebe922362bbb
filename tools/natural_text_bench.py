#!/usr/bin/env python3
"""Real natural-language text for BASELINE config 5 (enwik8 is not available offline): the prose that
ships with this image -- POD / reST / Markdown / README / licence files under /usr and /opt/rocm,
de-duplicated by content -- cut into documents of a given size, prepared and indexed on the device.
Prints build time, the path the suffix sort took, and (with --check) compares one document's tables
with the CPU oracle.

    python tools/natural_text_bench.py [--doc-mib 1] [--max-mib 24] [--keep-duplicates] [--check]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--doc-mib", type=float, default=1.0)
    ap.add_argument("--max-mib", type=float, default=24.0)
    ap.add_argument("--keep-duplicates", action="store_true", help="keep files with identical content (long repeats)")
    ap.add_argument("--check", action="store_true", help="compare document 0 with the CPU oracle")
    ap.add_argument("--resample-mib", type=float, default=0.0,
                    help="scale the corpus up: draw lines of the prose at random (with replacement) until this size")
    args = ap.parse_args()
    from east import hip_backend

    from east import synthetic
    raw, n_files = synthetic.image_prose(int(args.max_mib * (1 << 20)), args.keep_duplicates)
    if args.resample_mib > 0:
        rng = np.random.default_rng(20245)
        lines = [ln for ln in raw.split(b"\n") if len(ln) > 20]
        lens = np.array([len(ln) + 1 for ln in lines])
        picks = rng.integers(0, len(lines), size=int(args.resample_mib * (1 << 20) / lens.mean()) + 1)
        raw = b"\n".join(lines[i] for i in picks)[:int(args.resample_mib * (1 << 20))]
    step = int(args.doc_mib * (1 << 20))
    texts = [raw[i:i + step] for i in range(0, len(raw), step)]
    print("%d files, %.1f MiB of text in %d documents of %.2f MiB" % (n_files, len(raw) / 2**20, len(texts), args.doc_mib))

    hip_backend.unicode_tables()
    index = hip_backend.HipIndex()
    index.build_texts(texts)                   # warm-up: allocations
    t0 = time.perf_counter()
    index.build_texts(texts)
    wall = time.perf_counter() - t0
    info = index.info()
    sym, off, ms = index.prepared()
    print("symbols %d, strings %d, distinct text symbols %d, bits/symbol %d" % (info["n_total"], info["n_strings"],
                                                                               info["sigma_text"], info["bits_level0"]))
    print("build %.2f ms (%.2e chars/s), text preparation %.2f ms, wall %.1f ms; window sort %s, refinement rounds %d, "
          "DC3 levels %d, variable-length keys %d, fused finish %d, 64-bit passes %d"
          % (index.last_build_ms, len(raw) / (index.last_build_ms * 1e-3), index.last_prep_ms, wall * 1e3,
             "succeeded" if info["window_sorted"] else "gave up", info["refine_rounds"], info["dc3_levels"], info.get("ht_keys", 0),
             info["fused_finish"], info["radix_passes_u64"]))
    if os.environ.get("EAST_PROFILE"):
        index.profile_enable(True)
        index.build(sym, off, ms)
        report = index.profile_report()
        index.profile_enable(False)
        for name, (count, t_ms) in sorted(report.items(), key=lambda kv: -kv[1][1])[:22]:
            print("  %-36s %4d launches %8.3f ms" % (name, count, t_ms))
        print("  (build from resident symbols: %.2f ms)" % index.last_build_ms)
    lcp = index.tables(0, names=("lcptab",))["lcptab"]
    print("document 0: mean LCP %.1f, max LCP %d" % (float(lcp.mean()), int(lcp.max())))
    if args.check:
        from oracle import easa_oracle
        easa_oracle.build()
        d_sym = sym[off[0]:off[1]]
        o = easa_oracle.OracleEASA(symbols=d_sym, n_strings=int(ms[0]))
        t = index.tables(0)
        for name in ("suftab", "lcptab", "anntab", "childtab_up", "childtab_down", "childtab_next_l_index"):
            assert np.array_equal(t[name], getattr(o, name)), name
        print("document 0: all six tables bit-exact against the oracle")


if __name__ == "__main__":
    main()
