# -*- coding: utf-8 -*-
"""Enhanced annotated suffix array on MI355X -- the plugin behind AST.get_ast.

Replaces reference east/asts/easa.py (EnhancedAnnotatedSuffixArray, :12-36):
same constructor, attributes and score() surface; construction and the score
walk run in the HIP kernels behind include/east_hip.h.  Registered under
"easa_hip" and under the reference's algorithm names: the reference guarantees
(tests/asts/test_base.py:16-24) that "easa", "ast_linear" and "ast_naive"
give identical scores, so one backend serves all three names.
"""
import itertools

import numpy as np

from east import consts
from east import hip_backend
from east import utils as common_utils
from east.asts import base
from east.asts import utils


class HipEnhancedAnnotatedSuffixArray(base.AST):

    __algorithm__ = consts.ASTAlgorithm.EASA_HIP

    def __init__(self, strings_collection, device=None):
        super(HipEnhancedAnnotatedSuffixArray, self).__init__(strings_collection)   # empty check, base.py:20-22
        self.strings_collection = strings_collection
        symbols = utils.strings_to_symbols(strings_collection)                        # easa.py:19
        self._symbols = symbols
        self._index = hip_backend.HipIndex(device)
        self._index.build(symbols, np.array([0, symbols.size], dtype=np.int64),
                          np.array([len(strings_collection)], dtype=np.int32))       # easa.py:20-24
        self._tables = None

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
            synonyms = synonimizer.get_synonyms()
            query_words = common_utils.tokenize(query)
            for i in range(len(query_words)):
                query_words[i] = synonyms[query_words[i]] + [query_words[i]]
            possible_queries = ["".join(words) for words in itertools.product(*query_words)]
            return float(max(self._score_batch(possible_queries, normalized)))
        q = query.replace(" ", "")
        if not q:
            raise ZeroDivisionError("float division by zero")          # easa.py:134
        qs, qo = hip_backend.pack_queries([q])
        if return_suffix_scores:
            table, suf = self._index.score_table(qs, qo, normalized, want_suffix=True)
            return float(table[0, 0]), {q[i:]: float(suf[0, i]) for i in range(len(q))}
        return float(self._index.score_table(qs, qo, normalized)[0, 0])

    def _score_batch(self, queries, normalized=True):
        queries = [q.replace(" ", "") for q in queries]
        if not all(queries):
            raise ZeroDivisionError("float division by zero")
        qs, qo = hip_backend.pack_queries(queries)
        return self._index.score_table(qs, qo, normalized)[:, 0]

    # -- traversals over lcp-intervals (easa.py:38-89), host side, from the tables --
    def traverse_depth_first_pre_order(self, callback):
        n = len(self.suftab)
        stack = [[0, 0, n - 1, ""]]                                     # <l, i, j, char>
        while stack:
            interval = stack.pop()
            callback(interval)
            i, j = interval[1], interval[2]
            if i != j:
                children = self._get_child_intervals(i, j)
                children.sort(key=lambda child: child[3])
                stack.extend(reversed(children))

    def traverse_depth_first_post_order(self, callback):
        lcptab = self.lcptab
        last_interval = None
        n = len(lcptab)
        stack = [[0, 0, None, []]]                                      # <l, i, j, children>
        for i in range(1, n):
            lb = i - 1
            while lcptab[i] < stack[-1][0]:
                stack[-1][2] = i - 1
                last_interval = stack.pop()
                callback(last_interval)
                lb = last_interval[1]
                if lcptab[i] <= stack[-1][0]:
                    stack[-1][3].append(last_interval)
                    last_interval = None
            if lcptab[i] > stack[-1][0]:
                if last_interval:
                    stack.append([int(lcptab[i]), lb, None, [last_interval]])
                    last_interval = None
                else:
                    stack.append([int(lcptab[i]), lb, None, []])
        stack[-1][2] = n - 1
        callback(stack[-1])

    def traverse_breadth_first(self, callback):
        raise NotImplementedError                                       # easa.py:87-89

    def _lcp_value(self, i, j):                                         # easa.py:349-356
        n = len(self.suftab)
        if (i == 0 or i == n - 1) and j == n - 1:
            return 0
        up = int(self.childtab_up[j + 1])
        if i < up <= j:
            return int(self.lcptab[up])
        return int(self.lcptab[self.childtab_down[i]])

    def _get_child_intervals(self, i, j):                               # easa.py:358-377
        if i == j:
            return []
        n = len(self.suftab)
        sym, suftab, nxt = self._symbols, self.suftab, self.childtab_next_l_index
        l = self._lcp_value(i, j)
        ch = lambda r: chr(int(sym[suftab[r] + l]))
        intervals = []
        if i == 0 and j == n - 1:
            i1 = 0
        else:
            i1 = int(self.childtab_up[j + 1]) if i < self.childtab_up[j + 1] else int(self.childtab_down[i])
            intervals.append((self._lcp_value(i, i1 - 1), i, i1 - 1, ch(i)))
        while nxt[i1] != 0:
            i2 = int(nxt[i1])
            intervals.append((self._lcp_value(i1, i2 - 1), i1, i2 - 1, ch(i1)))
            i1 = i2
        intervals.append((self._lcp_value(i1, j), i1, j, ch(i1)))
        return intervals


class _EasaName(HipEnhancedAnnotatedSuffixArray):
    __algorithm__ = consts.ASTAlgorithm.EASA


class _LinearName(HipEnhancedAnnotatedSuffixArray):
    __algorithm__ = consts.ASTAlgorithm.AST_LINEAR


class _NaiveName(HipEnhancedAnnotatedSuffixArray):
    __algorithm__ = consts.ASTAlgorithm.AST_NAIVE
