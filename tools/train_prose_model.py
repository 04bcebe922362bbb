#!/usr/bin/env python3
"""Trains the character model of east/synthetic.py: prose_like_*: an order-3 model over 49 characters (lower-case letters,
space, newline, digits, common punctuation) from the prose that ships with the container image (POD / reST / Markdown /
licence texts: synthetic.image_prose -- enwik8 is not available offline), the 8 likeliest successors of every context kept.
Writes ast-text-analysis_amd/east/data/prose_order3.npz (committed: the generator must not depend on the image at hand)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
from east import synthetic      # noqa: E402

A, ORDER, TOP = synthetic.PROSE_ALPHABET_SIZE, 3, 8
raw, n_files = synthetic.image_prose(64 << 20, False)
text = raw.decode("utf-8", errors="replace").lower()
# (prose only: the image's documentation is tables of numbers, hex dumps and command lines as much as sentences)
lines = [ln for ln in text.split("\n") if len(ln) >= 30 and sum(c.isalpha() or c == " " for c in ln) >= 0.9 * len(ln)]
text = "\n".join(lines)
lut = np.full(0x110000, synthetic.PROSE_ALPHABET.index(" "), dtype=np.uint8)
for i, ch in enumerate(synthetic.PROSE_ALPHABET):
    lut[ord(ch)] = i
ids = lut[np.frombuffer(text.encode("utf-32-le"), dtype=np.uint32)]
# (runs of blanks / newlines collapse: the source is indented code examples and tables as much as prose)
keep = np.ones(ids.size, dtype=bool)
blank = (ids == synthetic.PROSE_ALPHABET.index(" ")) | (ids == synthetic.PROSE_ALPHABET.index("\n"))
keep[1:] = ~(blank[1:] & blank[:-1])
ids = ids[keep].astype(np.int64)
ctx = (ids[:-3] * A + ids[1:-2]) * A + ids[2:-1]
counts = np.bincount(ctx * A + ids[3:], minlength=A ** 4).reshape(A ** 3, A)
order = np.argsort(-counts, axis=1, kind="stable")[:, :TOP]
top = np.take_along_axis(counts, order, axis=1).astype(np.float64)
total = top.sum(axis=1, keepdims=True)
cum = np.where(total > 0, np.cumsum(top, axis=1) / np.maximum(total, 1), 0.0).astype(np.float32)
cum[total[:, 0] > 0, -1] = 1.0
starts = np.flatnonzero((total[:, 0] > 0) & (np.arange(A ** 3) // (A * A) == synthetic.PROSE_ALPHABET.index(" ")))
start_w = counts[starts].sum(axis=1).astype(np.float64)
seen = np.flatnonzero(total[:, 0] > 0)
out = os.path.join(ROOT, "ast-text-analysis_amd", "east", "data", "prose_order3.npz")
np.savez_compressed(out, contexts=seen.astype(np.int32), succ=order[seen].astype(np.uint8), cum=cum[seen],
                    starts=starts.astype(np.int32), start_cdf=(np.cumsum(start_w) / start_w.sum()).astype(np.float64))
print("trained on %d files, %d characters: %d contexts, %d start contexts -> %s (%d bytes)"
      % (n_files, ids.size, seen.size, starts.size, out, os.path.getsize(out)))
