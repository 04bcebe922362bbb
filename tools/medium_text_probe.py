#!/usr/bin/env python3
"""Natural-language-like documents of a quarter to four MiB -- the sizes `east keyphrases table` meets in practice --:
the device build with the refinement rounds launch by launch (east_hip_debug_set_lds_rounds(3)) and with the
persistent launch that finishes a domain that fits the chip (1, the default; csrc/persist_rounds.h).
    python tools/medium_text_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
import torch  # noqa: E402
from east import hip_backend, synthetic  # noqa: E402


def build_ms(sym, off, ms, knob):
    lib = hip_backend.load()
    lib.east_hip_debug_set_lds_rounds(knob)
    dev = torch.device("cuda", 0)
    d_sym = torch.from_numpy(sym.view(np.int32)).to(dev)
    index = hip_backend.HipIndex(0, reserve_symbols=int(sym.size))
    times = []
    for _ in range(6):
        index.build_device(d_sym.data_ptr(), sym.size, off, ms)
        times.append(index.last_build_ms)
    info = index.info()
    index.close()
    lib.east_hip_debug_set_lds_rounds(1)
    return min(times[1:]), info


def main():
    vocab = synthetic.zipf_vocabulary(np.random.default_rng(20245))
    for kind in ("zipf", "prose"):
        for size in (256 << 10, 1 << 20, 4 << 20):
            for n_docs in (1, 8):
                rng = np.random.default_rng(20240 + 9)
                if kind == "zipf":
                    docs = [synthetic.zipf_document(rng, size // n_docs, vocab) for _ in range(n_docs)]
                    sym = np.concatenate([d[0] for d in docs])
                    off = np.concatenate([[0], np.cumsum([d[0].size for d in docs])]).astype(np.int64)
                    ms = np.array([d[1] for d in docs], dtype=np.int32)
                else:
                    texts = synthetic.prose_like_texts(rng, n_docs, size // n_docs)
                    index = hip_backend.HipIndex(0)
                    index.build_texts(texts)
                    sym, off, ms = index.prepared()
                    index.close()
                old, info_old = build_ms(sym, off, ms, 3)
                new, info_new = build_ms(sym, off, ms, 1)
                print("%-5s %4d KiB in %d doc(s), %8d symbols: launch by launch %.3f ms (%d rounds), persistent %.3f ms (%d rounds, %d of them in one launch)"
                      % (kind, size >> 10, n_docs, sym.size, old, info_old["refine_rounds"], new, info_new["refine_rounds"], info_new["persist_rounds"]))


if __name__ == "__main__":
    main()
