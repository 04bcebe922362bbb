# -*- coding: utf-8 -*-
"""Names and values of the EAST option strings (the values a user passes to the reference:
east/consts.py:6-74).  Each group is a read-only namespace that can be iterated."""


class _Group(object):
    """Immutable bag of named string constants; iterating yields the values."""

    def __init__(self, **values):
        object.__setattr__(self, "_values", dict(values))
        for name, value in values.items():
            object.__setattr__(self, name, value)

    def __setattr__(self, name, value):
        raise AttributeError("constants are immutable")

    def __iter__(self):
        return iter(sorted(self._values.values(), key=str))

    def __contains__(self, value):
        return value in self._values.values()


# order in which AST.traverse() visits nodes (base.py:28-34)
TraversalOrder = _Group(DEPTH_FIRST_PRE_ORDER="depth-first|pre-order",
                        DEPTH_FIRST_POST_ORDER="depth-first|post-order",
                        BREADTH_FIRST="breadth-first")

# first code point of the per-string terminators (asts/utils.py:25-40)
String = _Group(UNICODE_SPECIAL_SYMBOLS_START=0x0A00)

# -s option of the CLI
RelevanceMeasure = _Group(AST="AST", COSINE="cosine")

# -a option of the CLI / second argument of AST.get_ast; "easa_hip" names the MI355X backend explicitly
ASTAlgorithm = _Group(EASA="easa", AST_LINEAR="ast_linear", AST_NAIVE="ast_naive", EASA_HIP="easa_hip")

# -w / -v options (cosine measure only; kept so that option parsing stays compatible)
TermWeighting = _Group(TF="tf", TF_IDF="tf-idf")
VectorSpace = _Group(WORDS="words", STEMS="stems", LEMMATA="lemmata")

# -l option
Language = _Group(**{name.upper(): name for name in (
    "danish", "dutch", "english", "finnish", "french", "german", "hungarian", "italian",
    "norwegian", "portuguese", "romanian", "russian", "spanish", "swedish")})
