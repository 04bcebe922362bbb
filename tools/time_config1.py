#!/usr/bin/env python3
"""Wall time of BASELINE config 1 (the reference's own CPU-runnable case: 30 HSE documents x 10
keyphrases, tests/golden/hse_config1.json) through the public API, keyphrases_table end to end."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
from east import applications, relevance  # noqa: E402

g = json.load(open(os.path.join(ROOT, "tests", "golden", "hse_config1.json")))
texts = {k: v.encode("utf-8") for k, v in g["texts"].items()}
measure = relevance.ASTRelevanceMeasure("easa", True)
applications.keyphrases_table(g["keyphrases"], texts, measure)       # warm-up: library load, allocations
times = []
for _ in range(20):
    t0 = time.perf_counter()
    table = applications.keyphrases_table(g["keyphrases"], texts, relevance.ASTRelevanceMeasure("easa", True))
    times.append(time.perf_counter() - t0)
total = sum(table[k][t] for k in table for t in table[k])
print("config 1: %d documents (%d bytes), %d keyphrases: keyphrases_table %.2f ms median (min %.2f), sum of scores %.15f"
      % (len(texts), sum(len(v) for v in texts.values()), len(g["keyphrases"]), sorted(times)[10] * 1e3, min(times) * 1e3, total))
if os.environ.get("EAST_PROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        applications.keyphrases_table(g["keyphrases"], texts, relevance.ASTRelevanceMeasure("easa", True))
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
if os.environ.get("EAST_BREAKDOWN"):
    from east import hip_backend, utils
    prepared = [utils.prepare_text(k) for k in g["keyphrases"]]
    qs, qo = hip_backend.pack_queries([p.replace(" ", "") for p in prepared])
    index = hip_backend.HipIndex()
    tl = list(texts.values())
    index.build_texts(tl)
    t = {"build_texts": [], "score_table": []}
    for _ in range(50):
        t0 = time.perf_counter(); index.build_texts(tl); t1 = time.perf_counter()
        index.score_table(qs, qo, True); t2 = time.perf_counter()
        t["build_texts"].append(t1 - t0); t["score_table"].append(t2 - t1)
    print("build_texts %.3f ms (device: prep %.3f + build %.3f), score_table %.3f ms (device %.3f)"
          % (sorted(t["build_texts"])[25] * 1e3, index.last_prep_ms, index.last_build_ms,
             sorted(t["score_table"])[25] * 1e3, index.last_score_ms))
    index.profile_enable(True)
    index.build_texts(tl)
    rep = index.profile_report()
    index.profile_enable(False)
    print("launches per build_texts:", sum(c for c, _ in rep.values()), sorted(rep.items(), key=lambda kv: -kv[1][0])[:8])
