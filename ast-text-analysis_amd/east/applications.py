# -*- coding: utf-8 -*-
"""keyphrases_table / keyphrases_graph (reference east/applications.py:11-149)."""
import itertools

from east import consts
from east import logging
from east import relevance
from east import utils


def keyphrases_table(keyphrases, texts, similarity_measure=None, synonimizer=None,
                     language=consts.Language.ENGLISH):
    """
    Constructs the keyphrases table, containing their matching scores in a set of texts.

    The resulting table is stored as a dictionary of dictionaries,
    where the entry table["keyphrase"]["text"] corresponds
    to the matching score (0 <= score <= 1) of keyphrase "keyphrase"
    in the text named "text".

    :param keyphrases: list of strings
    :param texts: dictionary of form {text_name: text}
    :param similarity_measure: similarity measure to use
    :param synonimizer: SynonymExtractor object to be used
    :param language: Language of the text collection / keyphrases

    :returns: dictionary of dictionaries, having keyphrases on its first level and texts
              on the second level.
    """
    similarity_measure = similarity_measure or relevance.ASTRelevanceMeasure()

    text_titles = list(texts.keys())
    text_collection = list(texts.values())
    similarity_measure.set_text_collection(text_collection, language)

    keyphrases_prepared = {keyphrase: utils.prepare_text(keyphrase) for keyphrase in keyphrases}
    res = {}

    # batched path: one score call for the whole table
    if synonimizer is None and hasattr(similarity_measure, "relevance_table"):
        wanted = [kp for kp in dict.fromkeys(keyphrases) if kp]          # applications.py:44-45
        if wanted:
            scores = similarity_measure.relevance_table([keyphrases_prepared[kp] for kp in wanted])
            for r, keyphrase in enumerate(wanted):
                res[keyphrase] = {text_titles[j]: float(scores[r, j]) for j in range(len(text_titles))}
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
    """
    Constructs the keyphrases relation graph based on the given texts corpus
    (reference east/applications.py:59-149): a keyphrase occurs in a text if its
    matching score reaches relevance_threshold; A -> B if at least
    referral_confidence of the texts containing A also contain B; nodes with
    fewer than support_threshold texts are dropped.

    :returns: {"nodes": [{"id", "label", "support"}], "edges": [{"source", "target", "confidence"}],
               "referral_confidence", "relevance_threshold", "support_threshold"}
    """
    similarity_measure = similarity_measure or relevance.ASTRelevanceMeasure()

    table = keyphrases_table(keyphrases, texts, similarity_measure, synonimizer, language)

    keyphrase_texts = {keyphrase: set([text for text in texts
                                       if table[keyphrase][text] >= relevance_threshold])
                       for keyphrase in keyphrases}

    graph = {
        "nodes": [
            {
                "id": i,
                "label": keyphrase,
                "support": len(keyphrase_texts[keyphrase])
            } for i, keyphrase in enumerate(keyphrases)
        ],
        "edges": [],
        "referral_confidence": referral_confidence,
        "relevance_threshold": relevance_threshold,
        "support_threshold": support_threshold
    }

    graph["nodes"] = [n for n in graph["nodes"]
                      if len(keyphrase_texts[n["label"]]) >= support_threshold]

    for i1, i2 in itertools.permutations(range(len(graph["nodes"])), 2):
        node1 = graph["nodes"][i1]
        node2 = graph["nodes"][i2]
        confidence = (float(len(keyphrase_texts[node1["label"]] &
                                keyphrase_texts[node2["label"]])) /
                      max(len(keyphrase_texts[node1["label"]]), 1))
        if confidence >= referral_confidence:
            graph["edges"].append({
                "source": node1["id"],
                "target": node2["id"],
                "confidence": confidence
            })

    return graph
