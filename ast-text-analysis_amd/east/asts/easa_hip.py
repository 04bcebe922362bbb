# -*- coding: utf-8 -*-
"""Enhanced annotated suffix array on MI355X -- the plugin behind AST.get_ast.

Replaces reference east/asts/easa.py (EnhancedAnnotatedSuffixArray, :12-36):
same constructor, attributes and score() surface; construction and the score
walk run in the HIP kernels behind include/east_hip.h.  Registered under
"easa_hip" and under the reference's algorithm names: the reference guarantees
(tests/asts/test_base.py:16-24) that "easa", "ast_linear" and "ast_naive"
give identical scores, so one backend serves all three names.
"""
import numpy as np

from east import consts
from east import hip_backend
from east import relevance
from east.asts import base
from east.asts import intervals
from east.asts import utils


class HipEnhancedAnnotatedSuffixArray(base.AST):

    __algorithm__ = consts.ASTAlgorithm.EASA_HIP

    def __init__(self, strings_collection, device=None):
        super(HipEnhancedAnnotatedSuffixArray, self).__init__(strings_collection)   # empty check, base.py:20-22
        self.strings_collection = strings_collection
        symbols = utils.strings_to_symbols(strings_collection)                        # easa.py:19
        self._symbols = utils.reference_code_points(symbols)      # the code points of the reference's self.string
        self._index = hip_backend.HipIndex(device)
        self._index.build(symbols, np.array([0, symbols.size], dtype=np.int64),
                          np.array([len(strings_collection)], dtype=np.int32))       # easa.py:20-24
        self._tables = None
        self._lefts = None

    # -- reference attributes (easa.py:18-24), fetched from the device on demand --
    @property
    def string(self):
        return "".join(chr(int(c)) if int(c) < 0x110000 else "�" for c in self._symbols)

    def _table(self, name):
        if self._tables is None:
            self._tables = self._index.tables(0)
        return self._tables[name]

    suftab = property(lambda self: self._table("suftab"))
    lcptab = property(lambda self: self._table("lcptab"))
    anntab = property(lambda self: self._table("anntab"))
    childtab_up = property(lambda self: self._table("childtab_up"))
    childtab_down = property(lambda self: self._table("childtab_down"))
    childtab_next_l_index = property(lambda self: self._table("childtab_next_l_index"))

    # -- score (easa.py:26-36) ---------------------------------------------------
    def score(self, query, normalized=True, synonimizer=None, return_suffix_scores=False):
        if synonimizer:
            # easa.py:27-34: the maximum over the variants, each scored with normalized=True whatever was asked for
            variants = relevance.synonym_variants(query, synonimizer)
            qs, qo = hip_backend.pack_queries(variants, keep_spaces=True)
            return float(self._index.score_table_grouped(qs, qo, [0, len(variants)], True)[0, 0])
        q = query.replace(" ", "")
        if not q:
            raise ZeroDivisionError("float division by zero")          # easa.py:134
        qs, qo = hip_backend.pack_queries([q])
        if return_suffix_scores:
            table, suf = self._index.score_table(qs, qo, normalized, want_suffix=True)
            return float(table[0, 0]), {q[i:]: float(suf[0, i]) for i in range(len(q))}
        return float(self._index.score_table(qs, qo, normalized)[0, 0])

    # -- traversals over lcp-intervals (easa.py:38-89) ----------------------------
    # The interval tree comes from closed forms over the device tables (east/asts/intervals.py):
    # first l-indices (anntab > 0), their left boundaries (east_hip_get_lcp_intervals) and widths.
    def _interval_inputs(self):
        if self._lefts is None:
            self._lefts = self._index.lcp_interval_lefts(0)
        return self.lcptab, self.anntab, self._lefts

    def traverse_depth_first_pre_order(self, callback):
        lcptab, anntab, left = self._interval_inputs()
        for visit in intervals.pre_order(lcptab, anntab, left, self.childtab_down, self.suftab, self._symbols):
            callback(visit)

    def traverse_depth_first_post_order(self, callback):
        for visit in intervals.post_order(*self._interval_inputs()):
            callback(visit)

    def traverse_breadth_first(self, callback):
        raise NotImplementedError                                       # easa.py:87-89


class _EasaName(HipEnhancedAnnotatedSuffixArray):
    __algorithm__ = consts.ASTAlgorithm.EASA


class _LinearName(HipEnhancedAnnotatedSuffixArray):
    __algorithm__ = consts.ASTAlgorithm.AST_LINEAR


class _NaiveName(HipEnhancedAnnotatedSuffixArray):
    __algorithm__ = consts.ASTAlgorithm.AST_NAIVE
