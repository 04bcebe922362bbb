# -*- coding: utf-8 -*-
"""keyphrases_table / keyphrases_graph (reference east/applications.py:11-149)."""

from collections.abc import Mapping

import numpy as np

from east import consts
from east import logging
from east import relevance
from east import utils


class _ScoreRow(Mapping):
    """One keyphrase's row of a ScoreTable: {text name: score}, read off the K x D array."""

    __slots__ = ("_table", "_k")

    def __init__(self, table, k):
        self._table, self._k = table, k

    def __getitem__(self, title):
        return float(self._table.scores[self._k, self._table._column[title]])

    def __iter__(self):
        return iter(self._table.text_titles)

    def __len__(self):
        return len(self._table.text_titles)


class ScoreTable(Mapping):
    """What keyphrases_table returns on the batched path: {raw keyphrase: {text name: score}} (applications.py:46-52) as a
    read-only mapping over the K x D score array itself.  At BASELINE configs[2] the table holds 2.56 M scores: a dict of
    dicts of Python floats costs seconds to make and a gigabyte to keep, the device fills the array in a millisecond.
    Compares equal to the dict of dicts with the same content; `scores`, `keyphrases` (row order) and `text_titles`
    (column order) are there for consumers that work on the array (east/formatting.py, keyphrases_graph)."""

    def __init__(self, keyphrases, text_titles, scores):
        self.keyphrases, self.text_titles, self.scores = list(keyphrases), list(text_titles), scores
        self._row = {kp: k for k, kp in enumerate(self.keyphrases)}
        self._column = {}
        for d, title in enumerate(self.text_titles):         # (as dict(zip(titles, row)): a repeated title keeps its last column)
            self._column[title] = d

    def __getitem__(self, keyphrase):
        return _ScoreRow(self, self._row[keyphrase])

    def __iter__(self):
        return iter(self.keyphrases)

    def __len__(self):
        return len(self.keyphrases)

    def to_dict(self):
        """The reference's own return type: a plain, mutable, JSON-serialisable dict of dicts of Python floats."""
        titles = self.text_titles
        return {kp: dict(zip(titles, row)) for kp, row in zip(self.keyphrases, np.asarray(self.scores, dtype=np.float64).tolist())}


# Tables below this many scores come back as the reference's plain dict of dicts (json.dumps, item assignment and
# isinstance(table, dict) work as they do there; 65 536 scores cost about 10 ms to box).  From here on the table is a
# ScoreTable over the array (to_dict() gives the plain form): at configs[2], 2.56 M scores, the dict costs 0.4 s and the
# device fills the array in a millisecond.
ARRAY_TABLE_MIN_SCORES = 1 << 16


def keyphrases_table(keyphrases, texts, similarity_measure=None, synonimizer=None,
                     language=consts.Language.ENGLISH):
    """Matching score of every keyphrase in every text (reference east/applications.py:11-56).

    :param keyphrases: raw keyphrase strings (empty ones are skipped, duplicates collapse)
    :param texts: {text name: text}
    :param similarity_measure: defaults to ASTRelevanceMeasure() (easa, normalized)
    :returns: {raw keyphrase: {text name: score}} -- a plain dict as in the reference; from ARRAY_TABLE_MIN_SCORES scores
              on, a ScoreTable (a read-only mapping over the score array that compares equal to that dict; `.to_dict()`)
    """
    similarity_measure = similarity_measure or relevance.ASTRelevanceMeasure()

    text_titles = list(texts.keys())
    text_collection = list(texts.values())
    similarity_measure.set_text_collection(text_collection, language)

    keyphrases_prepared = {keyphrase: utils.prepare_text(keyphrase) for keyphrase in keyphrases}
    res = {}

    # batched path: one score call for the whole table
    if hasattr(similarity_measure, "relevance_table"):
        wanted = [kp for kp in dict.fromkeys(keyphrases) if kp]          # applications.py:44-45
        if wanted:
            prepared = [keyphrases_prepared[kp] for kp in wanted]
            scores = (similarity_measure.relevance_table(prepared, synonimizer) if synonimizer
                      else similarity_measure.relevance_table(prepared))
            # (a mapping over the array: no K x D Python floats.  A rank of a multi-process run that is not the one to
            # print gets a K x 0 array -- east/parallel.py, table_rank --: rows without entries, as zip() made them)
            table = ScoreTable(wanted, text_titles[:scores.shape[1]], scores)
            return table if scores.size >= ARRAY_TABLE_MIN_SCORES else table.to_dict()
        return res

    i = 0
    total_scores = len(text_collection) * len(keyphrases)
    for keyphrase in keyphrases:
        if not keyphrase:
            continue
        res[keyphrase] = {}
        for j in range(len(text_collection)):
            i += 1
            logging.progress("Calculating matching scores", i, total_scores)
            res[keyphrase][text_titles[j]] = similarity_measure.relevance(
                keyphrases_prepared[keyphrase], text=j, synonimizer=synonimizer)
    logging.clear()
    return res


def keyphrases_graph(keyphrases, texts, referral_confidence=0.6, relevance_threshold=0.25,
                     support_threshold=1, similarity_measure=None, synonimizer=None,
                     language=consts.Language.ENGLISH):
    """Keyphrase implication graph over a text corpus (reference east/applications.py:59-149).

    A keyphrase *occurs* in a text when its matching score reaches `relevance_threshold`; its
    *support* is the number of such texts.  Keyphrases with support below `support_threshold` are
    dropped; for every ordered pair (A, B) of the remaining ones an edge A -> B is drawn when at
    least `referral_confidence` of the texts containing A also contain B.

    :returns: {"nodes": [{"id", "label", "support"}], "edges": [{"source", "target", "confidence"}],
               "referral_confidence", "relevance_threshold", "support_threshold"}; node ids are the
               positions of the keyphrases in the input list.
    """
    measure = similarity_measure or relevance.ASTRelevanceMeasure()
    table = keyphrases_table(keyphrases, texts, measure, synonimizer, language)

    if isinstance(table, ScoreTable) and len(set(table.text_titles)) == len(table.text_titles):
        return _graph_from_array(keyphrases, table, referral_confidence, relevance_threshold, support_threshold)

    occurs_in = {}
    if isinstance(table, ScoreTable):                        # (one comparison over the array instead of K x D look-ups)
        hits = table.scores >= relevance_threshold
        for keyphrase in keyphrases:
            found = set(table.text_titles[d] for d in np.flatnonzero(hits[table._row[keyphrase]]).tolist())
            occurs_in[keyphrase] = set(name for name in texts if name in found)
    else:
        for keyphrase in keyphrases:
            row = table[keyphrase]
            occurs_in[keyphrase] = set(name for name in texts if row[name] >= relevance_threshold)

    nodes = [{"id": position, "label": keyphrase, "support": len(occurs_in[keyphrase])}
             for position, keyphrase in enumerate(keyphrases)
             if len(occurs_in[keyphrase]) >= support_threshold]

    edges = []
    for source in nodes:
        source_texts = occurs_in[source["label"]]
        for target in nodes:
            if target is source:
                continue
            shared = len(source_texts & occurs_in[target["label"]])
            confidence = float(shared) / max(len(source_texts), 1)
            if confidence >= referral_confidence:
                edges.append({"source": source["id"], "target": target["id"], "confidence": confidence})

    return {"nodes": nodes, "edges": edges, "referral_confidence": referral_confidence,
            "relevance_threshold": relevance_threshold, "support_threshold": support_threshold}


def _graph_from_array(keyphrases, table, referral_confidence, relevance_threshold, support_threshold):
    """keyphrases_graph on the K x D score array: the same nodes and edges in the same order as the loops above (sources in
    the order of the keyphrase list, a source's targets in that order too), the pair counts by matrix products over
    blocks of sources instead of K^2 set intersections in Python (10 000 keyphrases: 10^8 of them)."""
    rows = np.array([table._row[kp] for kp in keyphrases], dtype=np.int64)            # (a repeated keyphrase: the same row twice)
    hits = table.scores[rows] >= relevance_threshold                                   # [n, D]
    support = hits.sum(axis=1)
    kept = np.flatnonzero(support >= support_threshold)
    nodes = [{"id": int(position), "label": keyphrases[position], "support": int(support[position])} for position in kept]
    edges = []
    h = hits[kept].astype(np.float32)                                                  # (counts up to D are exact in float32 below 2^24)
    sup = support[kept].astype(np.float64)
    block = max(1, (1 << 24) // max(len(kept), 1))
    for b in range(0, len(kept), block):
        shared = (h[b:b + block] @ h.T).astype(np.float64)                             # texts that hold source AND target
        confidence = shared / np.maximum(sup[b:b + block, None], 1.0)                  # float(shared) / max(len(source_texts), 1)
        src, dst = np.nonzero(confidence >= referral_confidence)
        off = src + b != dst                                                           # (no edge from a node to itself)
        for i, j, c in zip((src[off] + b).tolist(), dst[off].tolist(), confidence[src[off], dst[off]].tolist()):
            edges.append({"source": int(kept[i]), "target": int(kept[j]), "confidence": c})
    return {"nodes": nodes, "edges": edges, "referral_confidence": referral_confidence,
            "relevance_threshold": relevance_threshold, "support_threshold": support_threshold}
