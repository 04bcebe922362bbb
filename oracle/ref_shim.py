"""TEST INFRASTRUCTURE ONLY -- import shim for the (Python 2) reference.

Makes `import east` resolve to the *reference* package under
/root/reference/east, translated to Python 3 on the fly (lib2to3 + four
textual patches).  Nothing is copied to
disk: sources are read in place, refactored in memory and exec'd.

This module only works in the build container (where /root/reference is
mounted).  It is used by
  * oracle/gen_golden.py      -- to generate tests/golden/*.json
  * tests/test_oracle_vs_reference.py (auto-skipped when the reference is
    absent, e.g. on the GPU box)
and by nothing else.  The product package never imports it.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
import unittest
import warnings

REFERENCE_ROOT = os.environ.get("EAST_REFERENCE_ROOT", "/root/reference")
_PKG_DIR = os.path.join(REFERENCE_ROOT, "east")


def reference_available():
    return os.path.isfile(os.path.join(_PKG_DIR, "asts", "easa.py"))


def _refactor(src, path):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor
        fixers = refactor.get_fixers_from_package("lib2to3.fixes")
        tool = refactor.RefactoringTool(fixers)
        return str(tool.refactor_string(src + "\n", path))


def _patch(modname, src):
    """Semantic py2->py3 patches lib2to3 cannot know about."""
    if modname == "east.asts.easa":
        # (1) integer division inside _kark_sort (easa.py:156-158,180,182,203,205)
        head, sep, tail = src.partition("def _kark_sort")
        body, sep2, rest = tail.partition("def _radixpass")
        body = body.replace(" / ", " // ").replace("j/3", "j//3")
        src = head + sep + body + sep2 + rest
        # (2) np.int was removed from numpy (easa.py:150,256,276,277,296,312)
        src = src.replace("dtype=np.int)", "dtype=np.int64)")
    elif modname == "east.asts.utils":
        # (3) str has no .decode in py3 (asts/utils.py:39)
        src = src.replace('hex_code.decode("unicode-escape")',
                          'hex_code.encode("ascii").decode("unicode-escape")')
    elif modname == "east.utils":
        # (4) prepare_text: decode only bytes (utils.py:32)
        src = src.replace(
            "text = str(text.decode('utf-8', errors='replace'))",
            "text = text.decode('utf-8', errors='replace') if isinstance(text, bytes) else text")
    return src


class _RefFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != "east" and not fullname.startswith("east."):
            return None
        rel = fullname.split(".")[1:]
        base = os.path.join(_PKG_DIR, *rel)
        if os.path.isdir(base) and os.path.isfile(os.path.join(base, "__init__.py")):
            spec = importlib.machinery.ModuleSpec(fullname, self, is_package=True,
                                                  origin=os.path.join(base, "__init__.py"))
            spec.submodule_search_locations = [base]
            return spec
        if os.path.isfile(base + ".py"):
            return importlib.machinery.ModuleSpec(fullname, self, origin=base + ".py")
        return None

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        path = module.__spec__.origin
        with open(path, encoding="utf-8") as f:
            src = f.read()
        src = _patch(module.__name__, _refactor(src, path))
        module.__file__ = path
        exec(compile(src, path, "exec"), module.__dict__)


_installed = False


def install():
    """Install the finder + stubs; afterwards `import east` is the reference."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    if "east" in sys.modules and not getattr(sys.modules["east"], "__file__", "").startswith(REFERENCE_ROOT):
        raise RuntimeError("a different `east` package is already imported")
    # import-time-only dependencies of the reference that the image lacks
    nltk = types.ModuleType("nltk")
    corpus = types.ModuleType("nltk.corpus")
    corpus.stopwords = types.SimpleNamespace(words=lambda lang: [])
    stem = types.ModuleType("nltk.stem")
    snowball = types.ModuleType("nltk.stem.snowball")
    snowball.SnowballStemmer = lambda lang: types.SimpleNamespace(stem=lambda t: t)
    stem.snowball = snowball
    nltk.corpus, nltk.stem = corpus, stem
    sys.modules.setdefault("nltk", nltk)
    sys.modules.setdefault("nltk.corpus", corpus)
    sys.modules.setdefault("nltk.stem", stem)
    sys.modules.setdefault("nltk.stem.snowball", snowball)
    tt = types.ModuleType("testtools")
    tt.TestCase = unittest.TestCase
    sys.modules.setdefault("testtools", tt)
    sys.meta_path.insert(0, _RefFinder())
    _installed = True
