# -*- coding: utf-8 -*-
"""Text renderings of the keyphrase table and the keyphrase graph.

Byte-for-byte the outputs of the reference's east/formatting.py (table2xml, table2csv,
graph2gml, graph2edges) -- pinned by fixtures generated from it -- with two documented fixes:
format_table works (the reference refers to an undefined name) and graph2edges looks nodes up
by id (the reference indexes the node list, which breaks once a node was filtered out).
"""

_SCORE = "%.3f"


def format_table(table, format):
    renderers = {"xml": table2xml, "csv": table2csv}
    if format not in renderers:
        raise Exception("Unknown table format: '%s'. Please use one of: 'xml', 'csv'." % format)
    return renderers[format](table)


def _bulk(keyphrases_table, kind):
    """A ScoreTable (applications.py) of many scores goes through the library's host-side formatter -- the same bytes,
    written by a few threads straight from the score array -- instead of one Python '%.3f' per score.  None: not such a
    table (a plain dict, a small table, names or scores the formatter does not take)."""
    from east import applications
    if not isinstance(keyphrases_table, applications.ScoreTable) or keyphrases_table.scores.size < _BULK_MIN_SCORES:
        return None
    import numpy as np
    table = keyphrases_table
    scores = np.asarray(table.scores, dtype=np.float64)
    titles = table.text_titles
    if len(table) == 0 or len(set(titles)) != len(titles) or not np.isfinite(scores).all() or np.abs(scores).max() >= 1e15:
        return None
    try:
        from east import hip_backend
        kp_order = sorted(range(len(table.keyphrases)), key=table.keyphrases.__getitem__)
        text_order = sorted(range(len(titles)), key=titles.__getitem__)
        if kind == "xml":
            return hip_backend.format_table(scores, kp_order, text_order, table.keyphrases, titles, "xml")
        return hip_backend.format_table(scores, kp_order, text_order, [_csv_quote(k) for k in table.keyphrases],
                                        [_csv_quote(t) for t in titles], "csv")
    except (ValueError, UnicodeError, Exception) as e:       # noqa: BLE001 (the library is missing, a NUL in a name ...)
        from east import exceptions
        if isinstance(e, (ValueError, UnicodeError, exceptions.HipBackendError)):
            return None
        raise


_BULK_MIN_SCORES = 4096


def table2xml(keyphrases_table):
    """<table> / <keyphrase value=..> / <text name=..>score</text>, keyphrases and texts sorted."""
    text = _bulk(keyphrases_table, "xml")
    if text is not None:
        return text
    lines = ["<table>"]
    for keyphrase in sorted(keyphrases_table):
        row = keyphrases_table[keyphrase]
        lines.append('  <keyphrase value="%s">' % keyphrase)
        lines.extend('    <text name="%s">%s</text>' % (text, _SCORE % row[text]) for text in sorted(row))
        lines.append("  </keyphrase>")
    lines.append("</table>")
    return "\n".join(lines) + "\n"


def _csv_quote(value):
    return '"%s"' % value.replace('"', "'")


def table2csv(keyphrases_table):
    """Header row of quoted keyphrases, then one row per text: "name",score,score,..."""
    text = _bulk(keyphrases_table, "csv")
    if text is not None:
        return text
    keyphrases = sorted(keyphrases_table)
    texts = sorted(keyphrases_table[keyphrases[0]])
    rows = ["," + ",".join(_csv_quote(k) for k in keyphrases)]
    for text in texts:
        scores = [_SCORE % keyphrases_table[k][text] for k in keyphrases]
        rows.append(",".join([_csv_quote(text)] + scores))
    return "\n".join(rows) + "\n"


def format_graph(graph, format):
    renderers = {"gml": graph2gml, "edges": graph2edges}
    if format not in renderers:
        raise Exception("Unknown graph format: '%s'. Please use one of: 'gml', 'edges'." % format)
    return renderers[format](graph)


def graph2edges(graph):
    """One line per source node: `label -> label, label`, in order of first appearance."""
    label_of = dict((node["id"], node["label"]) for node in graph["nodes"])
    targets = {}
    for edge in graph["edges"]:
        targets.setdefault(label_of[edge["source"]], []).append(label_of[edge["target"]])
    return "".join("%s -> %s\n" % (source, ", ".join(found)) for source, found in targets.items())


def graph2gml(graph):
    """Graph Modelling Language: header with the three thresholds, a node block per node, an edge block per edge."""
    out = ["graph", "[", "  directed 1",
           "  referral_confidence %.2f" % graph["referral_confidence"],
           "  relevance_threshold %.2f" % graph["relevance_threshold"],
           "  support_threshold %i" % graph["support_threshold"]]
    for node in graph["nodes"]:
        out += ["  node", "  [", "    id %i" % node["id"], '    label "%s"' % node["label"], "  ]"]
    for edge in graph["edges"]:
        out += ["  edge", "  [", "    source %i" % edge["source"], "    target %i" % edge["target"],
                "    confidence %.2f" % edge["confidence"], "  ]"]
    out.append("]")
    return "\n".join(out) + "\n"
