# -*- coding: utf-8 -*-
"""Text preparation + plumbing (reference east/utils.py:31-133).

These functions define the input of the GPU path: a text becomes a list of
short strings (groups of three filtered tokens), a keyphrase becomes an
upper-cased query.  They stay on the host.
"""
import importlib
import itertools
import os
import re
import sys

_TOKEN_RE = re.compile(r"[\w']+", re.U)


def _upper_1to1(text):
    # The reference runs on Python 2, whose unicode.upper() maps code point to
    # code point ("ß" stays); Python 3's may expand ("ß" -> "SS").  Keep 1:1.
    up = text.upper()
    if len(up) == len(text):
        return up
    return "".join(c.upper() if len(c.upper()) == 1 else c for c in text)


def prepare_text(text):
    """utf-8 decode (errors='replace') + upper (utils.py:31-34)."""
    if isinstance(text, bytes):
        text = text.decode("utf-8", errors="replace")
    return _upper_1to1(text)


def tokenize(text):
    """utils.py:37-38."""
    return re.findall(_TOKEN_RE, text)


def text_to_strings_collection(text, words=3):
    """Split a text into strings of `words` consecutive tokens (utils.py:49-79):
    tokens of length <= 2 and all-digit tokens are dropped, groups are
    concatenated without separator, an empty result becomes [" "]."""
    text = prepare_text(text)
    tokens = [s for s in tokenize(text) if len(s) > 2 and not s.isdigit()]
    grouped = ["".join(tokens[i:i + words]) for i in range(0, len(tokens), words)]
    if not grouped:
        grouped = [" "]
    return grouped


def text_collection_to_string_collection(text_collection, words=3):
    """utils.py:82-83."""
    return flatten([text_to_strings_collection(text) for text in text_collection])


def flatten(lst):
    return list(itertools.chain.from_iterable(lst))


def output_is_redirected():
    """utils.py:96-97."""
    try:
        return os.fstat(0) != os.fstat(1)
    except OSError:
        return True


def itersubclasses(cls, _seen=None):
    """Generator over all subclasses of a class, depth first (utils.py:100-116)."""
    if not isinstance(cls, type):
        raise TypeError("itersubclasses must be called with new-style classes, not %.100r" % cls)
    _seen = _seen or set()
    for sub in cls.__subclasses__():
        if sub not in _seen:
            _seen.add(sub)
            yield sub
            for sub2 in itersubclasses(sub, _seen):
                yield sub2


def import_modules_from_package(package):
    """Import every module of a package so that its AST subclasses register with
    the factory (utils.py:119-133)."""
    pkg = importlib.import_module(package)
    for path in pkg.__path__:
        for filename in sorted(os.listdir(path)):
            if filename.startswith("__") or not filename.endswith(".py"):
                continue
            module_name = "%s.%s" % (package, filename[:-3])
            if module_name not in sys.modules:
                importlib.import_module(module_name)
