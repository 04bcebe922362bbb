# -*- coding: utf-8 -*-
"""Relevance measures (reference east/relevance.py:16-53, AST half).

ASTRelevanceMeasure keeps the reference surface (set_text_collection /
relevance) and adds the batched path the GPU needs: ONE build call for the whole
text collection (the documents become one device-resident shard of annotated
suffix arrays) and `relevance_table`, ONE score call for all keyphrases.
CosineRelevanceMeasure is a different method and out of scope.
"""
import itertools
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


def synonym_variants(query, synonimizer):
    """The queries easa.py:27-33 scores for a keyphrase under a synonimizer: every word of the
    (prepared) keyphrase may be replaced by one of its synonyms -- the product of the per-word
    alternatives, synonyms first, each variant the concatenation of its words.  `synonimizer`
    is anything with get_synonyms() -> {word: [synonyms]}; a word missing from the mapping
    raises KeyError exactly as the reference's dictionary look-up does."""
    alternatives = synonym_alternatives(query, synonimizer)
    if has_empty_variant(alternatives):
        raise ZeroDivisionError("float division by zero")              # easa.py:134 on an empty variant
    return ["".join(words) for words in itertools.product(*alternatives)]


def has_empty_variant(alternatives):
    """Whether the product of the per-word alternatives holds an empty string: no words at all, or an empty
    alternative for every word."""
    return all("" in alts for alts in alternatives)


def synonym_alternatives(query, synonimizer):
    """Per word of the prepared keyphrase: its synonyms followed by the word itself (what synonym_variants forms the
    product of).  Linear in the number of words: this is what the row cache is keyed by and what the multi-rank path
    validates on every rank.  Raises KeyError for a word missing from the mapping."""
    synonyms = synonimizer.get_synonyms()
    return tuple(tuple(synonyms[word]) + (word,) for word in utils.tokenize(query))


class _Shard(object):
    """The device index of a measure plus a one-row score cache.  Shared by the measure and its
    per-document views without a reference back to either, so that dropping the measure releases
    the device handle at once (no reference cycle waiting for the garbage collector)."""

    def __init__(self):
        self.index = None
        self.row_cache = (None, None, None)
        self.host_symbols = None

    def row(self, query, normalized, synonimizer=None):
        q = query.replace(" ", "")
        # (the synonym row is keyed by the per-word alternatives -- linear in the words, the product is only formed on a
        # miss: a synonimizer whose mapping changes, or a new one at a collected one's address, must not get a stale row)
        key = (q, bool(normalized)) if synonimizer is None else ("synonyms", synonym_alternatives(query, synonimizer))
        if self.row_cache[0] != key:
            if synonimizer is not None:
                # easa.py:27-34: max over the variants, scored with normalized=True whatever was asked for
                variants = synonym_variants(query, synonimizer)
                qs, qo = hip_backend.pack_queries(variants, keep_spaces=True)
                row = self.index.score_table_grouped(qs, qo, [0, len(variants)], True)[0]
            else:
                if not q:
                    raise ZeroDivisionError("float division by zero")          # easa.py:134
                qs, qo = hip_backend.pack_queries([q])
                row = self.index.score_table(qs, qo, normalized)[0]
            self.row_cache = (key, None, row)
        return self.row_cache[2]

    def symbols(self, doc):
        """The EASA string of one document as code points (host copy, fetched once per build)."""
        if self.host_symbols is None:
            self.host_symbols = ast_utils.reference_code_points(self.index.symbols())
        off = self.index.doc_offsets
        return self.host_symbols[int(off[doc]):int(off[doc + 1])]

    def suffix_scores(self, query, normalized, doc):
        q = query.replace(" ", "")
        if not q:
            raise ZeroDivisionError("float division by zero")                  # easa.py:134
        qs, qo = hip_backend.pack_queries([q])
        table, suf = self.index.score_table(qs, qo, normalized, want_suffix=True)
        return float(table[0, doc]), {q[i:]: float(suf[doc, i]) for i in range(len(q))}


class _DocumentAST(object):
    """What `measure.asts[i]` is in the reference: something with .score()."""

    def __init__(self, shard, doc):
        self._shard, self._doc = shard, doc

    def score(self, query, normalized=True, synonimizer=None, return_suffix_scores=False):
        if synonimizer:                                                         # easa.py:27-34
            return float(self._shard.row(query, normalized, synonimizer)[self._doc])
        if return_suffix_scores:                                                # easa.py:132-137
            return self._shard.suffix_scores(query, normalized, self._doc)
        return self._shard.row(query, normalized)[self._doc]

    # the rest of the AST surface (base.py:28-34), from this document's tables in the shard
    def traverse(self, callback, order=consts.TraversalOrder.DEPTH_FIRST_PRE_ORDER):
        from east.asts import intervals
        index, d = self._shard.index, self._doc
        t = index.tables(d, names=("suftab", "lcptab", "anntab", "childtab_down"))
        left = index.lcp_interval_lefts(d)
        if order == consts.TraversalOrder.DEPTH_FIRST_POST_ORDER:
            visits = intervals.post_order(t["lcptab"], t["anntab"], left)
        elif order == consts.TraversalOrder.DEPTH_FIRST_PRE_ORDER:
            visits = intervals.pre_order(t["lcptab"], t["anntab"], left, t["childtab_down"], t["suftab"],
                                         self._shard.symbols(d))
        else:
            raise NotImplementedError                                           # easa.py:87-89
        for visit in visits:
            callback(visit)


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
        texts = list(texts)
        self.texts = texts
        self.language = language
        if os.environ.get("EAST_HIP_TEXT_PREP", "device") == "device":
            # utils.text_to_strings_collection + make_unique_endings (relevance.py:44-45) on the device
            if self.index is None:
                self.index = hip_backend.HipIndex(self.device)
            self.index.build_texts(list(texts))
            self.asts = [_DocumentAST(self._shard, d) for d in range(len(texts))]
            self._shard.row_cache = (None, None, None)
            self._shard.host_symbols = None
            return
        collections = [utils.text_to_strings_collection(text) for text in texts]   # relevance.py:44-45
        self.set_strings_collections(collections)

    def set_strings_collections(self, collections):
        """collections[d] = the strings collection of document d (one AST each)."""
        parts = [ast_utils.strings_to_symbols(sc) for sc in collections]
        if any(ast_utils.is_tagged(p) for p in parts):       # text at or above U+0A00 somewhere: one encoding for the shard
            parts = [ast_utils.tag_terminators(p) for p in parts]
        self._build_from_parts(parts, collections)

    def _build_from_parts(self, parts, collections):
        doc_offsets = np.zeros(len(parts) + 1, dtype=np.int64)
        np.cumsum([p.size for p in parts], out=doc_offsets[1:])
        n_strings = np.array([len(sc) for sc in collections], dtype=np.int32)
        symbols = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
        if self.index is None:
            self.index = hip_backend.HipIndex(self.device)
        self.index.build(symbols, doc_offsets, n_strings)
        self.asts = [_DocumentAST(self._shard, d) for d in range(len(parts))]
        self._shard.row_cache = (None, None, None)
        self._shard.host_symbols = ast_utils.reference_code_points(symbols)

    def relevance(self, keyphrase, text, synonimizer=None):
        """relevance.py:51-53: the score of a prepared keyphrase in text number `text`."""
        return float(self._shard.row(keyphrase, self.normalized, synonimizer or None)[text])

    # HOT LOOP B (applications.py:43-52) as one batched call
    def relevance_table(self, prepared_keyphrases, synonimizer=None):
        """K prepared keyphrases -> K x D float64 array of scores.  With a synonimizer every
        keyphrase is expanded into its variants (synonym_variants), all variants of all keyphrases
        are scored in the one call and a segmented max on the device folds them back (easa.py:27-34)."""
        if synonimizer:
            groups = [synonym_variants(kp, synonimizer) for kp in prepared_keyphrases]
            offsets = np.zeros(len(groups) + 1, dtype=np.int64)
            np.cumsum([len(g) for g in groups], out=offsets[1:])
            qs, qo = hip_backend.pack_queries([v for g in groups for v in g], keep_spaces=True)
            return self.index.score_table_grouped(qs, qo, offsets, True)
        queries = [kp.replace(" ", "") for kp in prepared_keyphrases]
        if not all(queries):
            raise ZeroDivisionError("float division by zero")
        qs, qo = hip_backend.pack_queries(queries)
        return self.index.score_table(qs, qo, self.normalized)


class MultiDeviceASTRelevanceMeasure(ASTRelevanceMeasure):
    """ASTRelevanceMeasure over several GPUs of this process (`east -g N`): the documents are sharded over the devices --
    contiguous blocks balanced by size, one AST shard per device, no collective on the build path --, every shard scores
    its own documents and the K x D_local blocks are assembled by one all-gather (hip_backend.HipGroup ->
    east_hip_score_table_multi: RCCL between distinct devices).  Same surface and same numbers as the single-device
    measure (relevance.py:27-53); no torch, no child processes.  `devices`: a count or a list of device ordinals."""

    def __init__(self, ast_algorithm=consts.ASTAlgorithm.EASA, normalized=True, devices=1):
        super(MultiDeviceASTRelevanceMeasure, self).__init__(ast_algorithm, normalized, None)
        # (EAST_HIP_GROUP_DEVICES=0,0,1: the shards' device ordinals spelled out -- logical shards on one device, the tests)
        spelled = os.environ.get("EAST_HIP_GROUP_DEVICES", "")
        self.devices = [int(d) for d in spelled.split(",")] if spelled and isinstance(devices, int) else devices
        self.group = None
        self._shards = []

    def _after_build(self, n_docs):
        self._shards = []
        for view in self.group.shards:
            shard = _Shard()
            shard.index = view
            self._shards.append(shard)
        self.asts = []
        for d in range(n_docs):
            s = int(np.searchsorted(self.group.first_doc, d, side="right")) - 1
            self.asts.append(_DocumentAST(self._shards[s], int(d - self.group.first_doc[s])))
        self._row = (None, None)

    def set_text_collection(self, texts, language=consts.Language.ENGLISH):
        texts = list(texts)
        self.texts = texts
        self.language = language
        if self.group is None:
            self.group = hip_backend.HipGroup(self.devices)
        if os.environ.get("EAST_HIP_TEXT_PREP", "device") == "device":
            self.group.build_texts(texts)
            self._after_build(len(texts))
            return
        self.set_strings_collections([utils.text_to_strings_collection(text) for text in texts])

    def _build_from_parts(self, parts, collections):
        doc_offsets = np.zeros(len(parts) + 1, dtype=np.int64)
        np.cumsum([p.size for p in parts], out=doc_offsets[1:])
        n_strings = np.array([len(sc) for sc in collections], dtype=np.int32)
        if self.group is None:
            self.group = hip_backend.HipGroup(self.devices)
        self.group.build(np.concatenate(parts), doc_offsets, n_strings)
        self._after_build(len(parts))                        # (the shard views keep their slices of the symbols)

    def relevance(self, keyphrase, text, synonimizer=None):
        key = (keyphrase, bool(self.normalized), None if not synonimizer else synonym_alternatives(keyphrase, synonimizer))
        if self._row[0] != key:
            self._row = (key, self.relevance_table([keyphrase], synonimizer or None)[0])
        return float(self._row[1][text])

    def relevance_table(self, prepared_keyphrases, synonimizer=None):
        if synonimizer:
            # (the segmented max over the variants runs per shard: the blocks side by side are the table)
            groups = [synonym_variants(kp, synonimizer) for kp in prepared_keyphrases]
            offsets = np.zeros(len(groups) + 1, dtype=np.int64)
            np.cumsum([len(g) for g in groups], out=offsets[1:])
            qs, qo = hip_backend.pack_queries([v for g in groups for v in g], keep_spaces=True)
            blocks = [view.score_table_grouped(qs, qo, offsets, True) for view in self.group.shards if view.n_docs]
            return np.concatenate(blocks, axis=1)
        queries = [kp.replace(" ", "") for kp in prepared_keyphrases]
        if not all(queries):
            raise ZeroDivisionError("float division by zero")
        qs, qo = hip_backend.pack_queries(queries)
        return self.group.score_table(qs, qo, self.normalized)
