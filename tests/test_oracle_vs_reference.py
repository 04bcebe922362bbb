"""CPU tier, build container only: the oracle against the imported reference itself
(differential fuzz), plus the reference's own unit tests under the py3 shim.
Skipped where /root/reference is absent (the GPU box)."""
import random
import subprocess
import sys
import os

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_shim  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference not mounted here")

_SCRIPT = r'''
import sys, random
sys.path.insert(0, %(oracle)r); sys.path.insert(0, %(root)r)
import ref_shim; ref_shim.install()
from east.asts import base
import numpy as np
from oracle import easa_oracle as eo
rng = random.Random(%(seed)d)
n = 0
for alpha in ["AB", "ABC", "ABCDEFGH", "AB C", "ABCDEFGHIJKLMNOPQRSTUVWXYZ"]:
    for it in range(%(iters)d):
        m = rng.randint(1, 6)
        strings = ["".join(rng.choice(alpha) for _ in range(rng.randint(0, 12))) for _ in range(m)]
        if sum(map(len, strings)) + m < 2:
            continue
        ref = base.AST.get_ast(strings, "easa")
        lin = base.AST.get_ast(strings, "ast_linear")
        orc = eo.OracleEASA(strings)
        assert [ord(c) for c in ref.string] == orc.symbols.tolist()
        for name in ["suftab", "lcptab", "childtab_up", "childtab_down", "childtab_next_l_index", "anntab"]:
            assert np.array_equal(np.asarray(getattr(ref, name)), getattr(orc, name)), (name, strings)
        for _ in range(4):
            q = "".join(rng.choice(alpha + "Z") for _ in range(rng.randint(1, 10)))
            if not q.replace(" ", ""):
                continue
            for norm in (True, False):
                r = ref.score(q, normalized=norm, return_suffix_scores=True)
                assert float(lin.score(q, normalized=norm)) == float(r[0])
                for fast in (False, True):
                    o = orc.score(q, normalized=norm, return_suffix_scores=True, fast=fast)
                    assert float(r[0]) == o[0], (strings, q, norm, fast)
                    assert {k: float(v) for k, v in r[1].items()} == o[1]
        n += 1
print("OK", n)
'''


def _run(code):
    # a subprocess, because the shim makes `east` mean the reference package
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)


def test_oracle_equals_reference_on_fuzz():
    r = _run(_SCRIPT % {"oracle": os.path.join(ROOT, "oracle"), "root": ROOT, "seed": 77, "iters": 120})
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


def test_reference_own_unit_tests_pass_under_shim():
    code = (
        "import sys, unittest; sys.path.insert(0, %r); import ref_shim; ref_shim.install();"
        "sys.path.insert(0, %r);"
        "suite = unittest.defaultTestLoader.discover(%r, pattern='test_*.py', top_level_dir=%r);"
        "res = unittest.TextTestRunner(verbosity=0).run(suite);"
        "print('RAN', res.testsRun, len(res.failures) + len(res.errors))"
    ) % (os.path.join(ROOT, "oracle"), ref_shim.REFERENCE_ROOT,
         os.path.join(ref_shim.REFERENCE_ROOT, "tests"), ref_shim.REFERENCE_ROOT)
    r = _run(code)
    assert "RAN 7 0" in r.stdout, r.stdout + r.stderr


def test_golden_fixtures_are_reproducible(tmp_path):
    """gen_golden.py regenerates byte-identical fixtures (they really come from the reference)."""
    import filecmp, shutil
    gen = tmp_path / "oracle"
    shutil.copytree(os.path.join(ROOT, "oracle"), gen, ignore=shutil.ignore_patterns("*.so", "__pycache__"))
    (tmp_path / "tests" / "golden").mkdir(parents=True)
    # (the prose-like documents come out of the build's own generator: its module and its committed model, by path)
    east_dir = tmp_path / "ast-text-analysis_amd" / "east"
    (east_dir / "data").mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "ast-text-analysis_amd", "east", "synthetic.py"), str(east_dir / "synthetic.py"))
    shutil.copy(os.path.join(ROOT, "ast-text-analysis_amd", "east", "data", "prose_order3.npz"), str(east_dir / "data" / "prose_order3.npz"))
    r = subprocess.run([sys.executable, str(gen / "gen_golden.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    committed = sorted(f for f in os.listdir(os.path.join(ROOT, "tests", "golden")) if f.endswith(".json"))
    assert len(committed) == 11 and "high_text.json" in committed and "prose_like_docs.json" in committed and "traversal_synonyms.json" in committed
    assert sorted(os.listdir(str(tmp_path / "tests" / "golden"))) == committed      # every fixture, and nothing else
    for name in committed:
        assert filecmp.cmp(str(tmp_path / "tests" / "golden" / name), os.path.join(ROOT, "tests", "golden", name),
                           shallow=False), name
