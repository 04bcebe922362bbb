# -*- coding: utf-8 -*-
"""String enums of the EAST surface (names/values as in reference east/consts.py:6-74)."""


class _Enum(object):
    def __iter__(self):
        for k in dir(self):
            if not k.startswith("_"):
                yield getattr(self, k)

    def __setattr__(self, key, value):
        raise AttributeError("constants are immutable")


class _TraversalOrder(_Enum):
    DEPTH_FIRST_PRE_ORDER = "depth-first|pre-order"
    DEPTH_FIRST_POST_ORDER = "depth-first|post-order"
    BREADTH_FIRST = "breadth-first"


class _String(_Enum):
    UNICODE_SPECIAL_SYMBOLS_START = 0x0A00


class _RelevanceMeasure(_Enum):
    AST = "AST"
    COSINE = "cosine"


class _ASTAlgorithm(_Enum):
    AST_LINEAR = "ast_linear"
    AST_NAIVE = "ast_naive"
    EASA = "easa"
    EASA_HIP = "easa_hip"      # new: explicit name of the MI355X backend


class _TermWeighting(_Enum):
    TF = "tf"
    TF_IDF = "tf-idf"


class _VectorSpace(_Enum):
    WORDS = "words"
    STEMS = "stems"
    LEMMATA = "lemmata"


class _Language(_Enum):
    DANISH = "danish"
    DUTCH = "dutch"
    ENGLISH = "english"
    FINNISH = "finnish"
    FRENCH = "french"
    GERMAN = "german"
    HUNGARIAN = "hungarian"
    ITALIAN = "italian"
    NORWEGIAN = "norwegian"
    PORTUGUESE = "portuguese"
    ROMANIAN = "romanian"
    RUSSIAN = "russian"
    SPANISH = "spanish"
    SWEDISH = "swedish"


TraversalOrder = _TraversalOrder()
String = _String()
RelevanceMeasure = _RelevanceMeasure()
ASTAlgorithm = _ASTAlgorithm()
TermWeighting = _TermWeighting()
VectorSpace = _VectorSpace()
Language = _Language()
