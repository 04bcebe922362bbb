#!/usr/bin/env python3
"""Wall clock of the `east keyphrases table` CLI as a user runs it -- a fresh process per call -- on D files of S MiB
(prose-like text) and K keyphrases.  (Measured with it, round 5: 0.45-0.5 s whatever the collection -- interpreter start,
numpy, library load and device initialisation; starting the device on a background thread while the files are read changed
nothing, the files come out of the page cache in milliseconds -- that variant was taken out again.)
usage: cli_wall.py [D [S [K]]]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ast-text-analysis_amd")
sys.path.insert(0, PKG)
import numpy as np
from east import synthetic
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
tmp = tempfile.mkdtemp(prefix="east_cli_")
texts = synthetic.prose_like_texts(np.random.default_rng(1), D, int(S * (1 << 20)))
os.mkdir(os.path.join(tmp, "texts"))
for i, t in enumerate(texts):
    open(os.path.join(tmp, "texts", "doc%04d.txt" % i), "wb").write(t)
words = b" ".join(texts[:2]).split()
rng = np.random.default_rng(2)
with open(os.path.join(tmp, "kp.txt"), "wb") as f:
    for _ in range(K):
        st = int(rng.integers(0, len(words) - 3))
        f.write(b" ".join(words[st:st + int(rng.integers(1, 4))]) + b"\n")
env = dict(os.environ, PYTHONPATH=PKG)
for label, extra in (("east keyphrases table", {}),):
    walls = []
    for _ in range(4):
        t0 = time.perf_counter()
        out = subprocess.run([sys.executable, "-m", "east.main", "-f", "csv", "keyphrases", "table", os.path.join(tmp, "kp.txt"),
                              os.path.join(tmp, "texts")], env=dict(env, **extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        walls.append(time.perf_counter() - t0)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
    print("%-36s %d x %g MiB, %d keyphrases: process wall %s s (output %d bytes)" % (label, D, S, K, " ".join("%.3f" % w for w in walls), len(out.stdout)))
