#!/usr/bin/env python3
"""What the planning sample (csrc/east_hip.hip: sample_prefix_kernel) says about different kinds of text, next to
what the build then finds: first builds on fresh handles with EAST_HIP_TRACE=1 (the sample's counts go to stderr),
for every combination of the window / fused-finish knobs.  Used to set the thresholds in window_sort.h.

    python tools/plan_calibrate.py [--quick]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("EAST_HIP_TRACE", "1")


def corpora(quick):
    from east import synthetic
    rng = np.random.default_rng(20240)
    out = []
    _, s, m = synthetic.word_stream_document(rng, (8 if quick else 64) << 20, want_text=False)
    out.append(("words 1 doc", s, np.array([0, s.size]), np.array([m])))
    docs = [synthetic.word_stream_document(rng, 1 << 20, want_text=False)[1:] for _ in range(32 if quick else 256)]
    out.append(("words %d x 1 MiB" % len(docs), np.concatenate([d[0] for d in docs]),
                np.concatenate([[0], np.cumsum([d[0].size for d in docs])]), np.array([d[1] for d in docs])))
    vocab = synthetic.zipf_vocabulary(np.random.default_rng(20245))
    docs = [synthetic.zipf_document(rng, 1 << 20, vocab) for _ in range(16 if quick else 100)]
    out.append(("zipf %d x 1 MiB" % len(docs), np.concatenate([d[0] for d in docs]),
                np.concatenate([[0], np.cumsum([d[0].size for d in docs])]), np.array([d[1] for d in docs])))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    from east import hip_backend, synthetic
    cases = corpora(args.quick)
    raw, _ = synthetic.image_prose(24 << 20, False)
    if len(raw) > (2 << 20):
        rng = np.random.default_rng(20245)
        lines = [ln for ln in raw.split(b"\n") if len(ln) > 20]
        lens = np.array([len(ln) + 1 for ln in lines])
        mib = 16 if args.quick else 64
        picks = rng.integers(0, len(lines), size=int(mib * (1 << 20) / lens.mean()) + 1)
        big = b"\n".join(lines[i] for i in picks)[:mib << 20]
        hip_backend.unicode_tables()
        for name, texts in (("prose %d x 1 MiB (resampled)" % mib, [big[i:i + (1 << 20)] for i in range(0, len(big), 1 << 20)]),
                            ("prose as it is, 1 MiB docs", [raw[i:i + (1 << 20)] for i in range(0, len(raw), 1 << 20)])):
            prep = hip_backend.HipIndex()
            prep.build_texts(texts)
            s, off, ms = prep.prepared()
            cases.append((name, s, off, ms))
            prep.close()
    lib = hip_backend.load()
    for name, s, off, ms in cases:
        print("==== %s: %d symbols" % (name, s.size), flush=True)
        for knob, env_window, label in ((1, None, "default"), (4, None, "no fused finish"), (1, "wide", "wide window"),
                                        (4, "wide", "wide window, no fused finish")):
            lib.east_hip_debug_set_window_sort(knob)
            if env_window:
                os.environ["EAST_HIP_WINDOW"] = "12"
            else:
                os.environ.pop("EAST_HIP_WINDOW", None)
            sys.stderr.flush()
            best = None
            for rep in range(2):                        # fresh handles: first builds (the second one without the first-touch costs)
                index = hip_backend.HipIndex(reserve_symbols=int(s.size))
                index.build(s, off, ms)
                info = index.info()
                best = index.last_build_ms if best is None else min(best, index.last_build_ms)
                index.close()
            print("  %-30s build %7.3f ms  fused %d  rounds %d  kept %.3f  u64 passes %d" % (
                label, best, info["fused_finish"], info["refine_rounds"], info["first_kept"] / max(1, info["first_n"]),
                info["radix_passes_u64"]), flush=True)
        lib.east_hip_debug_set_window_sort(1)
        os.environ.pop("EAST_HIP_WINDOW", None)


if __name__ == "__main__":
    main()
