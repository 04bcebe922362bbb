# -*- coding: utf-8 -*-
"""Table output formats (reference east/formatting.py:4-39, with the documented
behaviour: the reference's format_table references an undefined name)."""


def format_table(table, format):
    if format == "xml":
        return table2xml(table)
    elif format == "csv":
        return table2csv(table)
    else:
        raise Exception("Unknown table format: '%s'. "
                        "Please use one of: 'xml', 'csv'." % format)


def table2xml(keyphrases_table):
    res = "<table>\n"
    for keyphrase in sorted(keyphrases_table.keys()):
        res += '  <keyphrase value="%s">\n' % keyphrase
        for text in sorted(keyphrases_table[keyphrase].keys()):
            res += '    <text name="%s">' % text
            res += '%.3f' % keyphrases_table[keyphrase][text]
            res += '</text>\n'
        res += '  </keyphrase>\n'
    res += "</table>\n"
    return res


def table2csv(keyphrases_table):

    def quote(s):
        return '"' + s.replace('"', "'") + '"'

    keyphrases = sorted(keyphrases_table.keys())
    texts = sorted(keyphrases_table[keyphrases[0]].keys())
    res = "," + ",".join(map(quote, keyphrases)) + "\n"  # Heading
    for text in texts:
        scores = ["%.3f" % keyphrases_table[keyphrase][text] for keyphrase in keyphrases]
        res += (quote(text) + "," + ",".join(scores) + "\n")
    return res
