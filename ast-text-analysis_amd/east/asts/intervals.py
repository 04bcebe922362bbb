# -*- coding: utf-8 -*-
"""lcp-interval tree of an enhanced suffix array, from closed forms over the tables.

The reference walks the tree with childtab look-ups (east/asts/easa.py:38-85,
349-377).  Here the tree is read off the arrays the device produces:

* an lcp-interval l-[i..j] with l > 0 is named by its first l-index k -- the rank
  with anntab[k] > 0 -- and l = lcptab[k], i = PSV(k), j = i + anntab[k] - 1
  (`left` = PSV comes from east_hip_get_lcp_intervals,
  the same pyramid search the annotation kernel uses);
* post-order = the intervals sorted by (right boundary, then inner before outer),
  the root 0-[0..n-1] last; an interval's children are the maximal intervals
  strictly inside it;
* pre-order adds the leaves: the ranks of an interval no child interval covers.

Pure numpy / Python, no device access: east/asts/easa_hip.py feeds it.
"""
import numpy as np


def internal_nodes(lcptab, anntab, left):
    """(l, i, j) int64 arrays of the non-root lcp-intervals in post-order."""
    k = np.flatnonzero(np.asarray(left) >= 0)
    i = np.asarray(left, dtype=np.int64)[k]
    j = i + np.asarray(anntab, dtype=np.int64)[k] - 1
    l = np.asarray(lcptab, dtype=np.int64)[k]
    order = np.lexsort((-i, j))                  # by j; intervals ending at the same rank: inner (larger i) first
    return l[order], i[order], j[order]


def post_order(lcptab, anntab, left):
    """Yields <l, i, j, children> lists as easa.py:57-85 hands them to its callback: every
    lcp-interval after all intervals inside it, children = the nested lists of its child
    intervals (internal ones only), the root last."""
    n = len(lcptab)
    l, i, j = internal_nodes(lcptab, anntab, left)
    pending = []                                 # finished intervals that have not met their parent yet
    for node in zip(l.tolist() + [0], i.tolist() + [0], j.tolist() + [n - 1]):
        children = []
        while pending and pending[-1][1] >= node[1]:
            children.append(pending.pop())
        children.reverse()
        visit = [node[0], node[1], node[2], children]
        yield visit
        pending.append(visit)


def leaf_depths(lcptab, childtab_down):
    """The l the reference reports for a singleton child [r..r] (easa.py:349-356 applied to
    i == j): lcptab[childtab_down[r]], and 0 for the last rank."""
    lcptab = np.asarray(lcptab, dtype=np.int64)
    out = lcptab[np.asarray(childtab_down, dtype=np.int64)]
    out[-1] = 0
    return out


def pre_order(lcptab, anntab, left, childtab_down, suftab, symbols):
    """Yields the visits of easa.py:38-55: the root [0, 0, n-1, ""], then every child as a tuple
    (l, i, j, char) -- char = the symbol that leads into it, string[suftab[i] + l(parent)] --
    depth first, children in rank order (= sorted by char), leaves included."""
    n = len(lcptab)
    tree = None
    for tree in post_order(lcptab, anntab, left):
        pass                                     # the last visit is the root with the whole tree nested inside
    leaf_l = leaf_depths(lcptab, childtab_down).tolist()
    suftab = np.asarray(suftab, dtype=np.int64)
    yield [0, 0, n - 1, ""]
    if n == 1:
        return
    # explicit stack of (interval, next rank to emit, index of the next child interval)
    stack = [[tree, 0, 0]]
    while stack:
        frame = stack[-1]
        (l, _, j, kids), r, c = frame[0], frame[1], frame[2]
        if r > j:
            stack.pop()
            continue
        ch = chr(int(symbols[suftab[r] + l]))
        if c < len(kids) and kids[c][1] == r:
            kid = kids[c]
            frame[1], frame[2] = kid[2] + 1, c + 1
            yield (kid[0], kid[1], kid[2], ch)
            stack.append([kid, kid[1], 0])
        else:
            frame[1] = r + 1
            yield (leaf_l[r], r, r, ch)
