#!/usr/bin/env python3
"""BUILD CONTAINER ONLY -- re-time the reference's own Python.

Imports the reference in place from /root/reference through oracle/ref_shim.py (lib2to3 + 4 patches; nothing is
copied) and times `base.AST.get_ast(strings, alg)` and `ast.score(query)` for easa and ast_linear on uniform
random A-Z collections of about 10^4, 10^5 and 10^6 symbols (fixed seeds), one core.  Prints the rows of
BASELINE.md section 2 as a Markdown table.  The reference never travels to the GPU box: this script refuses to
run where /root/reference is absent.

    python tools/time_reference.py [--max-n 1000000] [--queries 200]
"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_shim  # noqa: E402


def collection(rng, n_symbols):
    """3-word strings of uniform A-Z words of 3..10 letters, as `east keyphrases table` chunks a text."""
    strings, total = [], 0
    while total < n_symbols:
        s = "".join("".join(rng.choice("ABCDEFGHIJKLMNOPQRSTUVWXYZ") for _ in range(rng.randint(3, 10))) for _ in range(3))
        strings.append(s)
        total += len(s) + 1
    return strings, total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-n", type=int, default=1000000)
    ap.add_argument("--queries", type=int, default=200)
    args = ap.parse_args()
    if not ref_shim.reference_available():
        raise SystemExit("the reference is not mounted here (/root/reference): this script only runs in the build container")
    ref_shim.install()
    from east.asts import base
    print("| metric | value | input | entry point |")
    print("|---|---|---|---|")
    for n in (10000, 100000, 1000000):
        if n > args.max_n:
            break
        rng = random.Random(20240 + n)
        strings, total = collection(rng, n)
        queries = ["".join(rng.choice("ABCDEFGHIJKLMNOPQRSTUVWXYZ") for _ in range(rng.randint(8, 24))) for _ in range(args.queries)]
        for alg in ("easa", "ast_linear"):
            t0 = time.perf_counter()
            ast = base.AST.get_ast(strings, alg)
            t_build = time.perf_counter() - t0
            t0 = time.perf_counter()
            for q in queries:
                ast.score(q)
            t_score = time.perf_counter() - t0
            print("| %s build | %.2e symbols/s (%.2f s) | n = %d symbols, m = %d strings | `get_ast(sc, \"%s\")` |"
                  % (alg, total / t_build, t_build, total, len(strings), alg))
            print("| %s score | %.0f keyphrase-scores/s | %d queries of 8-24 letters vs that collection | `ast.score(q)` |"
                  % (alg, len(queries) / t_score, len(queries)))
            sys.stdout.flush()
    print("\n(one core of %d; CPython %s)" % (os.cpu_count(), sys.version.split()[0]))


if __name__ == "__main__":
    main()
