# -*- coding: utf-8 -*-
"""`east` CLI, table branch (reference east/main.py:15-122 as documented in
README.rst:27-63; the shipped reference main crashes with a NameError).

    east [-s ast] [-a easa|easa_hip|ast_linear|ast_naive] [-d] [-f xml|csv] \\
         keyphrases table <keyphrases file> <directory with .txt files | single file>
    east [-c confidence] [-r relevance] [-p support] [-f edges|gml] keyphrases graph <keyphrases file> <texts>

Several GPUs (one process per GPU, documents sharded over the ranks, one RCCL all-gather of the score blocks --
east/parallel.py; the reference's loops relevance.py:41-53 / applications.py:43-52 are what is sharded):
    east -g 8 keyphrases table ...          or   EAST_HIP_DEVICES=8 east keyphrases table ...
        starts the 8 ranks itself (child processes under torch.distributed.run) and relays rank 0's output;
    python -m torch.distributed.run --nproc-per-node 8 ... -m east.main keyphrases table ...
        under a launcher (WORLD_SIZE > 1 in the environment) every rank runs this module, rank 0 prints.
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


_OPTIONS = "s:a:w:v:l:f:c:r:p:g:dy"
_VALUE_OPTIONS = frozenset(c for c, nxt in zip(_OPTIONS, _OPTIONS[1:] + " ") if nxt == ":")


def _world():
    """(world size, rank) a launcher gave this process (torch.distributed.run exports both)."""
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))


def launch_ranks(n_ranks, argv):
    """The no-launcher spelling of a multi-GPU run (`-g N` / EAST_HIP_DEVICES=N): start N ranks of this module as
    CHILD processes under torch.distributed.run (east/launch.py), let their output through (only rank 0 prints) and
    return their exit code.  Nothing in this process has touched a GPU at this point, and it never exec()s.
    EAST_HIP_LAUNCHER replaces the launcher command (tests)."""
    from east import launch
    pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (the ranks must not start ranks of their own, and every rank takes the device of its LOCAL_RANK)
    return launch.run_ranks(n_ranks, ["-m", "east.main"] + list(argv), "EAST_HIP_LAUNCHER",
                            env_drop=("EAST_HIP_DEVICES", "EAST_HIP_DEVICE"), pythonpath=pkg)


def main(argv=None, measure_factory=None):
    """measure_factory: test hook of the multi-rank path (the per-shard scorer of DistributedASTRelevanceMeasure;
    the collective logic then runs on gloo without a GPU)."""
    args = sys.argv[1:] if argv is None else list(argv)
    try:
        opts, args = getopt.getopt(args, _OPTIONS)
    except getopt.GetoptError as e:
        print(e)
        return 1
    opt_list = opts                                         # (as given: repeated options, empty values)
    opts = dict(opts)
    world, rank = _world()
    if world == 1:
        try:
            n_ranks = int(opts.get("-g", os.environ.get("EAST_HIP_DEVICES", "1")))
        except ValueError:
            print("Invalid number of GPUs: '%s'." % opts.get("-g", os.environ.get("EAST_HIP_DEVICES")))
            return 1
        if n_ranks > 1 and os.environ.get("EAST_HIP_MULTI", "threads") != "process":
            # the default: N devices in THIS process (one AST shard and one host thread of the library per device, one
            # RCCL all-gather of the score blocks -- east_hip_score_table_multi); no child processes, no torch
            opts["-g"] = str(n_ranks)
        elif n_ranks > 1:
            # EAST_HIP_MULTI=process: one process per GPU under torch.distributed.run --
            # the same command line without -g: the ranks learn the world size from the launcher
            child_argv = []
            for key, value in opt_list:
                if key != "-g":
                    child_argv.extend([key, value] if key[1:] in _VALUE_OPTIONS else [key])
            child_argv.extend(args)
            return launch_ranks(n_ranks, child_argv)
    quiet = rank != 0                                       # under a launcher only rank 0 prints
    opts.setdefault("-l", consts.Language.ENGLISH)
    opts.setdefault("-s", consts.RelevanceMeasure.AST)
    opts.setdefault("-a", consts.ASTAlgorithm.EASA)

    saved_stdout = sys.stdout
    if quiet:
        sys.stdout = open(os.devnull, "w")
    try:
        return _main(opts, args, world, measure_factory)
    finally:
        if quiet:
            sys.stdout.close()
            sys.stdout = saved_stdout


def _main(opts, args, world, measure_factory):
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
    distributed = world > 1 or os.environ.get("EAST_HIP_FORCE_DIST") == "1"
    texts = {}
    if len(text_files) == 1:      # a single file: one text per line
        lines = _read(text_files[0]).splitlines()
        for i in range(len(lines)):
            texts[str(i)] = lines[i]
    elif distributed:
        # one rank per GPU: the documents are sharded by their FILE SIZES (os.stat); a rank reads its own files only
        from east import parallel
        for filename in text_files:
            texts[os.path.basename(filename)[:-4]] = parallel.LazyText(filename, _read)
    else:
        for filename in text_files:
            texts[os.path.basename(filename)[:-4]] = _read(filename)

    measure_name = opts["-s"]
    if measure_name.lower() != "ast":
        print("Relevance measure '%s' is not available in the MI355X build (only 'ast')." % measure_name)
        return 1
    if "-y" in opts:
        print("Synonym extraction (-y) needs the external Tomita parser and is not available.")
        return 1

    group_up = False
    try:
        if distributed:                                                   # (forced: the collective path with one rank)
            # one rank per GPU: this rank indexes its share of the documents on the device of its LOCAL_RANK and
            # the K x D table is assembled by one all-gather (east/parallel.py); every rank computes, rank 0 prints
            import torch.distributed as dist
            from east import parallel
            backend = os.environ.get("EAST_HIP_DIST_BACKEND", "nccl")      # (gloo: the CPU-tier tests)
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                kwargs = {}
                if backend == "nccl":
                    import torch
                    local = int(os.environ.get("LOCAL_RANK", "0"))
                    torch.cuda.set_device(local)
                    kwargs["device_id"] = torch.device("cuda", local)
                dist.init_process_group(backend, **kwargs)
                group_up = True
            device = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else None
            similarity_measure = parallel.DistributedASTRelevanceMeasure(opts["-a"], "-d" not in opts, device=device,
                                                                         measure_factory=measure_factory, table_rank=0)
        elif world == 1 and int(opts.get("-g", "1")) > 1:
            similarity_measure = relevance.MultiDeviceASTRelevanceMeasure(opts["-a"], "-d" not in opts, int(opts["-g"]))
        else:
            similarity_measure = relevance.ASTRelevanceMeasure(opts["-a"], "-d" not in opts)      # main.py:95-98
        return _run(subcommand, keyphrases, texts, similarity_measure, opts)
    except exceptions.EastException as e:       # (no device, a failed build ...): a message, not a traceback
        print(e)
        return 1
    finally:
        if group_up:
            import torch.distributed as dist
            dist.destroy_process_group()


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
