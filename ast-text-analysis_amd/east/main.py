# -*- coding: utf-8 -*-
"""`east` CLI, table branch (reference east/main.py:15-122 as documented in
README.rst:27-63; the shipped reference main crashes with a NameError).

    east [-s ast] [-a easa|easa_hip|ast_linear|ast_naive] [-d] [-f xml|csv] \\
         keyphrases table <keyphrases file> <directory with .txt files | single file>
    east [-c confidence] [-r relevance] [-p support] [-f edges|gml] keyphrases graph <keyphrases file> <texts>
"""
import getopt
import os
import sys

from east import applications
from east import consts
from east import exceptions
from east import formatting
from east import relevance


def _read(path):
    with open(path, "rb") as f:
        return f.read()


def main(argv=None):
    args = sys.argv[1:] if argv is None else list(argv)
    try:
        opts, args = getopt.getopt(args, "s:a:w:v:l:f:c:r:p:dy")
    except getopt.GetoptError as e:
        print(e)
        return 1
    opts = dict(opts)
    opts.setdefault("-l", consts.Language.ENGLISH)
    opts.setdefault("-s", consts.RelevanceMeasure.AST)
    opts.setdefault("-a", consts.ASTAlgorithm.EASA)

    if len(args) < 2:
        print("Invalid syntax: EAST should be called as:\n\n"
              "    east [options] <command> <subcommand> args\n\n"
              "Commands available: keyphrases.\n"
              "Subcommands available: table/graph.")
        return 1

    command, subcommand = args[0], args[1]
    if command != "keyphrases":
        print("Invalid command: '%s'. Please use one of: 'keyphrases'." % command)
        return 1
    if len(args) < 4:
        print('Invalid syntax. For keyphrases analysis, EAST should be called as:\n\n'
              '    east [options] keyphrases <subcommand> "path/to/keyphrases.txt" '
              '"path/to/texts/dir"')
        return 1

    keyphrases = _read(os.path.abspath(args[2])).decode("utf-8", errors="replace").splitlines()   # main.py:60-64

    text_collection_path = os.path.abspath(args[3])                                             # main.py:67-89
    if os.path.isdir(text_collection_path):
        text_files = [os.path.join(text_collection_path, filename)
                      for filename in sorted(os.listdir(text_collection_path)) if filename.endswith(".txt")]
    else:
        text_files = [text_collection_path]
    texts = {}
    if len(text_files) == 1:      # a single file: one text per line
        lines = _read(text_files[0]).splitlines()
        for i in range(len(lines)):
            texts[str(i)] = lines[i]
    else:
        for filename in text_files:
            texts[os.path.basename(filename)[:-4]] = _read(filename)

    measure_name = opts["-s"]
    if measure_name.lower() == "ast":
        similarity_measure = relevance.ASTRelevanceMeasure(opts["-a"], "-d" not in opts)          # main.py:95-98
    else:
        print("Relevance measure '%s' is not available in the MI355X build (only 'ast')." % measure_name)
        return 1
    if "-y" in opts:
        print("Synonym extraction (-y) needs the external Tomita parser and is not available.")
        return 1

    try:
        return _run(subcommand, keyphrases, texts, similarity_measure, opts)
    except exceptions.EastException as e:       # (no device, a failed build ...): a message, not a traceback
        print(e)
        return 1


def _run(subcommand, keyphrases, texts, similarity_measure, opts):
    if subcommand == "table":
        table = applications.keyphrases_table(keyphrases, texts, similarity_measure, None, opts["-l"])
        table_format = opts.get("-f", "xml").lower()
        try:
            print(formatting.format_table(table, table_format))
        except Exception as e:
            print(e)
            return 1
        return 0
    elif subcommand == "graph":                                                                 # main.py:124-143
        opts.setdefault("-c", "0.6")    # referral confidence
        opts.setdefault("-r", "0.25")   # relevance threshold of the matching score
        opts.setdefault("-p", "1")      # support threshold for graph nodes
        graph = applications.keyphrases_graph(keyphrases, texts, float(opts["-c"]), float(opts["-r"]),
                                              float(opts["-p"]), similarity_measure, None, opts["-l"])
        graph_format = opts.get("-f", "edges").lower()
        try:
            print(formatting.format_graph(graph, graph_format))
        except Exception as e:
            print(e)
            return 1
        return 0
    print("Invalid subcommand: '%s'. Please use one of: 'table', 'graph'." % subcommand)
    return 1


if __name__ == "__main__":
    sys.exit(main())
