#!/usr/bin/env python3
"""The drop-in call at BASELINE configs[2] scale, end to end: applications.keyphrases_table + formatting.format_table for
256 documents of 1 MiB and 10 000 keyphrases (2.56 M scores) -- what `east keyphrases table` runs (reference
east/main.py:67-122 -> applications.py:11-56 -> formatting.py:14-39) --, split into the device's part, the copy of the
table to the host, and the Python around them.  usage: time_config2_cli.py [docs [keyphrases]]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
from east import applications, formatting, hip_backend, relevance, synthetic, utils      # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
rng = np.random.default_rng(20240 + 3)
texts, syms = {}, []
for d in range(D):
    text, sym, _ = synthetic.word_stream_document(rng, 1 << 20)
    texts["doc%04d" % d] = text
    syms.append(sym)
qs, qo = synthetic.keyphrases(rng, np.concatenate(syms), K)
del syms
keyphrases = []
for k in range(K):                                           # back to text: words of at most 10 letters
    q = "".join(chr(int(c)) for c in qs[qo[k]:qo[k + 1]])
    keyphrases.append(" ".join(q[i:i + 10] for i in range(0, len(q), 10)))
hip_backend.unicode_tables()


def run(label, table_factory):
    t = {}
    measure = relevance.ASTRelevanceMeasure("easa", True)
    t0 = time.perf_counter()
    table = table_factory(measure)
    t["keyphrases_table"] = time.perf_counter() - t0
    index = measure.index
    t["  device: text preparation + build"] = (index.last_prep_ms + index.last_build_ms) * 1e-3
    t["  device: score kernels"] = index.last_score_ms * 1e-3
    t0 = time.perf_counter()
    xml = formatting.format_table(table, "xml")
    t["format_table xml (%d MB)" % (len(xml) >> 20)] = time.perf_counter() - t0
    t0 = time.perf_counter()
    csv = formatting.format_table(table, "csv")
    t["format_table csv (%d MB)" % (len(csv) >> 20)] = time.perf_counter() - t0
    print(label)
    for name, s in t.items():
        print("  %-44s %9.1f ms" % (name, s * 1e3))
    return xml, csv


def as_rounds_1_to_4(measure):
    """keyphrases_table as it was: the K x D array turned into a dict of dicts of Python floats"""
    table = applications.keyphrases_table(keyphrases, texts, measure)
    return {kp: dict(row.items()) for kp, row in table.items()}


run("warm-up", lambda m: applications.keyphrases_table(keyphrases[:100], texts, m))
a = run("dict of dicts + one '%.3f' per score (rounds 1-4)", as_rounds_1_to_4)
b = run("ScoreTable (a mapping over the array) + the library's formatter", lambda m: applications.keyphrases_table(keyphrases, texts, m))
print("identical output:", a == b)
t0 = time.perf_counter()
prepared = [utils.prepare_text(k) for k in keyphrases]
print("  (prepare_text of the keyphrases alone: %.1f ms)" % ((time.perf_counter() - t0) * 1e3))
