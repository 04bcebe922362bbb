#!/usr/bin/env python3
"""Extended differential run of the device text preparation (east_hip_build_texts) against the
host chain (prepare_text / tokenize / text_to_strings_collection / make_unique_endings), with the
checker of tests/test_gpu_parity.py: random Unicode from many scripts (whole ranges below and above
U+0A00, combining marks, digits of several scripts, case-special letters) and malformed UTF-8
(random byte junk, truncated and overlong sequences, surrogates), documents from empty to 50 kB.

    python tools/fuzz_textprep.py [--seconds 120] [--seed 1]
"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_gpu_parity as T  # noqa: E402
from east import hip_backend  # noqa: E402

RANGES = [(0x20, 0x7F), (0xA0, 0x17F), (0x180, 0x24F), (0x250, 0x2FF), (0x300, 0x36F), (0x370, 0x3FF), (0x400, 0x52F),
          (0x530, 0x58F), (0x590, 0x5FF), (0x600, 0x6FF), (0x700, 0x7FF), (0x900, 0x9FF), (0xA00, 0xA7F), (0x1E00, 0x1EFF),
          (0x1F00, 0x1FFF), (0x2000, 0x206F), (0x2070, 0x209F), (0x2150, 0x218F), (0x2460, 0x24FF), (0x3040, 0x309F),
          (0xFB00, 0xFB06), (0xFF10, 0xFF5A), (0x10400, 0x1044F), (0x1D7CE, 0x1D7FF), (0x1F600, 0x1F64F)]


def random_text(rng, max_len):
    out = []
    n = rng.choice([0, 1, 5, 40, 400, max_len])
    style = rng.random()                # (a third of the texts draw from every range: word characters at or above U+0A00)
    while sum(len(x) for x in out) < n:
        r = rng.random()
        if r < 0.08:
            out.append(bytes(rng.randrange(256) for _ in range(rng.randint(1, 4))))          # raw junk
        elif r < 0.12:
            cp = rng.choice([0xD800, 0xDFFF, 0x110000, 0x1FFFFF])                            # surrogates / too large
            out.append(bytes([0xF0 | (cp >> 18) & 7, 0x80 | (cp >> 12) & 63, 0x80 | (cp >> 6) & 63, 0x80 | cp & 63])
                       if cp > 0xFFFF else bytes([0xE0 | cp >> 12, 0x80 | (cp >> 6) & 63, 0x80 | cp & 63]))
        else:
            lo, hi = rng.choice(RANGES[:12] if style < 0.67 else RANGES)
            word = "".join(chr(rng.randint(lo, hi)) for _ in range(rng.randint(1, 9)))
            word = "".join(c for c in word if not 0xD800 <= ord(c) <= 0xDFFF)
            b = word.encode("utf-8")
            if rng.random() < 0.05 and len(b) > 1:
                b = b[:-1]                                                                    # truncated sequence
            out.append(b + rng.choice([b" ", b" ", b", ", b"\n", b"'", b"_", b"", b"12 "]))
    return b"".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = random.Random(args.seed)
    hip_backend.load()
    t_end = time.time() + args.seconds
    cases = texts_n = total = tagged = 0
    while time.time() < t_end:
        texts = [random_text(rng, rng.choice([200, 5000, 50000])) for _ in range(rng.randint(1, 8))]
        # (how the text goes up: in one piece, or in chunks of a few dozen bytes to tens of kilobytes -- the streamed preparation)
        hip_backend.load().east_hip_debug_set_text_stream(rng.choice([0, 0, 17, 64, 700, 9000, 40000]))
        index = T._check_device_prep(hip_backend, texts)
        tagged += hip_backend.load().east_hip_prepared_encoding(index._h)
        index.close()
        cases += 1
        texts_n += len(texts)
        total += sum(len(t) for t in texts)
    hip_backend.load().east_hip_debug_set_text_stream(-1)
    print("text preparation fuzz ok: %d collections (%d of them with kept text at or above U+0A00: tagged encoding), "
          "%d texts, %d bytes" % (cases, tagged, texts_n, total))


if __name__ == "__main__":
    main()
