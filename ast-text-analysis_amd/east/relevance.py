# -*- coding: utf-8 -*-
"""Relevance measures (reference east/relevance.py:16-53, AST half).

ASTRelevanceMeasure keeps the reference surface (set_text_collection /
relevance) and adds the batched path the GPU needs: ONE build call for the whole
text collection (the documents become one device-resident shard of annotated
suffix arrays) and `relevance_table`, ONE score call for all keyphrases.
CosineRelevanceMeasure is a different method and out of scope.
"""
import os

import numpy as np

from east import consts
from east import exceptions
from east import hip_backend
from east import utils
from east.asts import utils as ast_utils


class RelevanceMeasure(object):

    def set_text_collection(self, texts, language=consts.Language.ENGLISH):
        raise NotImplementedError()

    def relevance(self, keyphrase, text, synonimizer=None):
        # text is the index of the text to measure the relevance to
        raise NotImplementedError()


class _Shard(object):
    """The device index of a measure plus a one-row score cache.  Shared by the measure and its
    per-document views without a reference back to either, so that dropping the measure releases
    the device handle at once (no reference cycle waiting for the garbage collector)."""

    def __init__(self):
        self.index = None
        self.row_cache = (None, None, None)

    def row(self, query, normalized):
        q = query.replace(" ", "")
        if self.row_cache[0] != q or self.row_cache[1] != bool(normalized):
            if not q:
                raise ZeroDivisionError("float division by zero")              # easa.py:134
            qs, qo = hip_backend.pack_queries([q])
            self.row_cache = (q, bool(normalized), self.index.score_table(qs, qo, normalized)[0])
        return self.row_cache[2]


class _DocumentAST(object):
    """What `measure.asts[i]` is in the reference: something with .score()."""

    def __init__(self, shard, doc):
        self._shard, self._doc = shard, doc

    def score(self, query, normalized=True, synonimizer=None, return_suffix_scores=False):
        if synonimizer or return_suffix_scores:
            raise NotImplementedError("use east.asts.base.AST.get_ast(...) for synonym / per-suffix scoring")
        return self._shard.row(query, normalized)[self._doc]


class ASTRelevanceMeasure(RelevanceMeasure):

    def __init__(self, ast_algorithm=consts.ASTAlgorithm.EASA, normalized=True, device=None):
        super(ASTRelevanceMeasure, self).__init__()
        if ast_algorithm not in list(consts.ASTAlgorithm):
            from east import exceptions
            raise exceptions.NoSuchASTAlgorithm(name=ast_algorithm)
        self.ast_algorithm = ast_algorithm
        self.normalized = normalized
        self.device = device
        self._shard = _Shard()

    @property
    def index(self):
        return self._shard.index

    @index.setter
    def index(self, value):
        self._shard.index = value

    # HOT LOOP A (relevance.py:34-49) as one batched build
    def set_text_collection(self, texts, language=consts.Language.ENGLISH):
        self.texts = texts
        self.language = language
        if os.environ.get("EAST_HIP_TEXT_PREP", "device") == "device":
            # utils.text_to_strings_collection + make_unique_endings (relevance.py:44-45) on the device
            if self.index is None:
                self.index = hip_backend.HipIndex(self.device)
            try:
                self.index.build_texts(list(texts))
            except exceptions.HipBackendError as e:
                if "outside the method's domain" in str(e):
                    code = int(str(e).split("U+")[1].split()[0], 16)
                    raise exceptions.SymbolOutOfDomainException(code=code)
                raise
            self.asts = [_DocumentAST(self._shard, d) for d in range(len(texts))]
            self._shard.row_cache = (None, None, None)
            return
        collections = [utils.text_to_strings_collection(text) for text in texts]   # relevance.py:44-45
        self.set_strings_collections(collections)

    def set_strings_collections(self, collections):
        """collections[d] = the strings collection of document d (one AST each)."""
        parts = [ast_utils.strings_to_symbols(sc) for sc in collections]
        doc_offsets = np.zeros(len(parts) + 1, dtype=np.int64)
        np.cumsum([p.size for p in parts], out=doc_offsets[1:])
        n_strings = np.array([len(sc) for sc in collections], dtype=np.int32)
        symbols = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
        if self.index is None:
            self.index = hip_backend.HipIndex(self.device)
        self.index.build(symbols, doc_offsets, n_strings)
        self.asts = [_DocumentAST(self._shard, d) for d in range(len(parts))]
        self._shard.row_cache = (None, None, None)

    def _row(self, query, normalized):
        return self._shard.row(query, normalized)

    def relevance(self, keyphrase, text, synonimizer=None):
        """relevance.py:51-53: the score of a prepared keyphrase in text number `text`."""
        if synonimizer:
            raise NotImplementedError("synonym-expanded scoring is not part of the HIP hot path")
        return float(self._row(keyphrase, self.normalized)[text])

    # HOT LOOP B (applications.py:43-52) as one batched call
    def relevance_table(self, prepared_keyphrases):
        """K prepared keyphrases -> K x D float64 array of scores."""
        queries = [kp.replace(" ", "") for kp in prepared_keyphrases]
        if not all(queries):
            raise ZeroDivisionError("float division by zero")
        qs, qo = hip_backend.pack_queries(queries)
        return self.index.score_table(qs, qo, self.normalized)
