#!/usr/bin/env python3
"""bench.py -- the EAST hot path on N MI355X GPUs (one process per GPU).

Workloads (BASELINE.json):
  N = 1   configs[1]: one synthetic 64 MiB random-ASCII word-stream document, turned into 3-word strings
          exactly as `east keyphrases table` does (text mode), and 1 000 keyphrases -- the headline.  The
          same JSON line carries a second leg, `config2` = configs[2] (256 x 1 MiB documents, 10 000
          keyphrases: the keyphrase x document score kernel), which is also the per-GPU shape of configs[3],
          and a third, `config5` = 100 x 1 MiB natural-language-like (Zipf) documents, 1 000 keyphrases: the
          input class whose build goes through the tie-refinement rounds.
  N > 1   configs[3] (2 048 documents of 1 MiB over 8 GPUs): every rank owns 256 x 1 MiB documents (one AST
          shard per GPU, weak scaling), 10 000 keyphrases; the K x D_local score blocks (20.5 MB per rank)
          are assembled with one RCCL all-gather.  `--docs/--doc-mib/--keyphrases` override either shape.

A "step" is one pass of the whole hot path over that batch, with the symbol stream already resident in HBM:
    EASA build (dense remap, window sort / DC3 suffix array, LCP, annotation table; the child tables are
    built on the first east_hip_get_tables request that asks for them -- `child_tables_ms` reports that kernel)
    + score table (K keyphrases x local documents) [+ all-gather of the blocks]
value = input document bytes of all ranks / step time.

The JSON line also carries
  roofline       the dominant kernel (by summed HIP-event time on the library's own stream, found in an untimed
                 pass that brackets every kernel): its average launch duration measured live over the TIMED
                 steps -- there only its launches are bracketed, two events per launch cost the stream time --,
                 algorithmic bytes / time, PMC traffic from profiles/traffic.json;
                 roofline.rocprof_hbm_fraction = the whole build: sum over its kernels of (PMC bytes per
                 launch x launches) / sum of their HIP-event times, against the 8 TB/s peak
  roofline_score the score walk: 8 B per table read / binary-search probe (counted by the kernel in an
                 extra, untimed run) + 8 B per suffix result written and read + 8 B per score, / its time
  build_from_host_ms  east_hip_build from host-resident symbols (H2D included), wall clock
  build_ms_without_guesses  the build as a handle's first build runs it (alphabet and tie-group read-backs in place)
  cpu_baseline   the CPU oracle (C port of the reference's easa.py) timed on this box's host cores on the
                 same 64 MiB document (one core) and over the documents of configs[2] (all cores)

Launch:  python bench.py [--gpus N --steps K --warmup W]      (N > 1 without WORLD_SIZE in the environment: bench.py
                                                               starts its N ranks itself, as child processes)
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ast-text-analysis_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured)
BUILD_KERNEL_PREFIXES = ("radix_", "lvl0_", "ann_", "pyramid_", "presence_", "remap_", "codemap_", "validate_", "scan_",
                         "dc3_", "lcp", "doc_", "inverse_sa", "spec_counts", "refine_", "lg_", "sample_")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--doc-mib", type=float, default=None, help="size of one document in MiB (default: 64 at N=1, 1 at N>1)")
    ap.add_argument("--docs", type=int, default=None, help="documents per GPU (default: 1 at N=1, 256 at N>1)")
    ap.add_argument("--keyphrases", type=int, default=None, help="default: 1000 at N=1, 10000 at N>1")
    ap.add_argument("--mode", choices=["text", "direct"], default="text",
                    help="text: 3-word strings as the CLI does; direct: get_ast([one string])")
    ap.add_argument("--corpus", choices=["words", "zipf"], default="words",
                    help="words: uniform A-Z word stream (configs 1-3); zipf: natural-language-like (config 5)")
    ap.add_argument("--duplicate-docs", type=int, default=0,
                    help="the last N documents are copies of the first N (long repeats across documents)")
    ap.add_argument("--denormalized", action="store_true", help="the CLI's -d")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config2", action="store_true", help="skip the configs[2] and config 5 legs of the N=1 line")
    ap.add_argument("--no-extras", action="store_true", help="skip build_from_host / child tables / probe count")
    ap.add_argument("--base-value", type=float, default=float(os.environ.get("EAST_BENCH_BASE_VALUE", "0") or 0),
                    help="N > 1: the same-shape single-GPU value (config2.value of the N = 1 line, chars/s) -- the line "
                         "then carries multi_gpu.scaling_efficiency = value / (N x base)")
    ap.add_argument("--full-line", action="store_true",
                    help="print the whole record (what bench_detail.json holds) as the last line instead of the driver's compact "
                         "one: the A/B scripts under tools/ read per-kernel tables off it")
    ap.add_argument("--no-in-process", action="store_true",
                    help="N > 1: skip the leg that runs the same shards through the in-process device group on rank 0")
    ap.add_argument("--cpu-sample-mib", type=float, default=64.0,
                    help="size of the CPU-baseline document (64 = the bench document itself, about 15 s of one core)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))      # (never returns: the ranks are child processes)
    return args


def launch_ranks(n_ranks, argv):
    """`python bench.py --gpus N` (N > 1) without a launcher around it: start the N ranks ourselves, as CHILD
    processes under torch.distributed.run (east/launch.py: rendezvous on a port the launcher binds itself), relay what
    they print (rank 0's JSON line) and return their exit code.  Runs before torch is imported or anything else could
    touch a GPU -- this process never initialises one and never exec()s (a process that has initialised the GPU must
    not be replaced on this pool).  EAST_BENCH_LAUNCHER replaces the launcher command (tests: a stub that records its
    command line)."""
    from east import launch
    return launch.run_ranks(n_ranks, [os.path.abspath(__file__)] + list(argv), "EAST_BENCH_LAUNCHER")


def kernel_bytes(name, info, n, n_docs):
    """Algorithmic HBM bytes of ALL launches of one kernel in one step."""
    e32, e64 = info["radix_elements_u32"], info["radix_elements_u64"]
    first = n if info["window_sorted"] else 2 * n // 3      # elements of the level-0 window sort
    table = {
        # first radix pass: the window keys are generated from the byte stream (1 B/symbol), pairs written once
        "radix_scatter_kernel<u32,gen>": first * 9, "radix_scatter_kernel<u64,gen>": first * 13,
        "radix_hist_kernel<u32,gen>": first, "radix_hist_kernel<u64,gen>": first,
        # read key+value, write key+value
        "radix_scatter_kernel<u64>": e64 * 24, "radix_scatter_kernel<u32>": e32 * 16,
        # read keys once
        "radix_hist_kernel<u64>": e64 * 8, "radix_hist_kernel<u32>": e32 * 4,
        # per merged suffix: 4 B sorted-list read + 24 B (symbol, rank) window gather + 4 B write
        "dc3_merge_tile_kernel": info["merge_elements"] * 32,
        # fused merge + LCP: 4 B list read + one 16 B record gather + 4 B SA + 4 B LCP written per suffix
        "dc3_merge_lcp_tile_kernel": n * 28,
        # 4 B SA read, 2 x 16 B symbol windows, 4 B LCP write per rank (first comparison step)
        "lcp_kernel": n * 40,
        # byte stream: 4 B SA read, 2 x 8 B symbol windows, 4 B LCP write per rank
        "lcp8_kernel": n * 24,
        # level-0 placement pass (no refinement rounds): 4 B element + 4 B key read, 4 B SA (+ 4 B LCP with one
        # document) written per suffix; the tied ones add two 8 B text gathers each
        "lvl0_place_kernel": first * (16 if n_docs == 1 and info["window_sorted"] else 12),
        # the fused finish (last radix digit in LDS + placement): the pair read once (4 B key + 4 B element), 4 B SA + 4 B LCP
        # written per suffix; the tied ones add one 4 B text gather each
        "lvl0_finish_kernel": first * 16,
        # the in-LDS refinement round (all launches of a build: lds_sorted elements): element, group flag and slot read
        # (12 B), one 16-byte text gather, suffix + next domain's element + flag written (12 B)
        "refine_lds_sort_kernel": info.get("lds_sorted", 0) * 40,
        # 4 B LCP read + 4 B annotation write per rank + pyramid level 1 (4 B per 16 ranks)
        "ann_stream_kernel": n * 8 + n // 4,
        # 4 B read per symbol
        "presence_kernel": n * 4,
        # 4 B read + 1 B written per symbol
        "remap_bytes_kernel": n * 5,
        "presence_remap_kernel": n * 5,
        # 4 B sorted sample read + one random 4 B rank store per sample
        "dc3_rank_kernel": info["merge_elements"] * 2 // 3 * 8,
        # 8 symbols + 2 ranks read, one 16 B record written per symbol
        "dc3_records_kernel": n * 26,
    }
    return table.get(name)


def make_corpus(args, synthetic, rng, n_docs, doc_bytes):
    parts, ms = [], []
    vocab = synthetic.zipf_vocabulary(np.random.default_rng(20245)) if args.corpus == "zipf" else None
    for _ in range(n_docs):
        if vocab is not None:
            sym_d, m_d = synthetic.zipf_document(rng, doc_bytes, vocab)
        elif args.mode == "text":
            _, sym_d, m_d = synthetic.word_stream_document(rng, doc_bytes, want_text=False)
        else:
            sym_d, m_d = synthetic.direct_document(rng, doc_bytes + 1)
        parts.append(sym_d)
        ms.append(m_d)
    for i in range(min(args.duplicate_docs, len(parts) // 2)):
        parts[len(parts) - 1 - i], ms[len(ms) - 1 - i] = parts[i], ms[i]
    symbols = np.concatenate(parts) if len(parts) > 1 else parts[0]
    doc_offsets = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.int64)
    return symbols, doc_offsets, np.array(ms, dtype=np.int32), vocab is not None


def load_traffic(name, meta=None):
    """HBM bytes per launch per kernel from the PMC passes (profiles/traffic*.json, tools/summarize_profiles.py).
    meta (a dict) receives the file's `_commit`: the commit of the tree the passes were taken on."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        data = json.load(f)
    if meta is not None:
        meta["commit"] = data.get("_commit")
    return {k: v for k, v in data.items() if not k.startswith("_")}


def pmc_by_kernel(prof, steps, traffic, top=14):
    """Per kernel: time per step and -- where the PMC passes have a figure -- HBM bytes per step and the fraction of the
    8 TB/s peak that is (counter bytes / HIP-event time)."""
    total = sum(ms for _, ms in prof.values())
    rows = []
    for name, (launches, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:top]:
        row = {"kernel": name, "ms_per_step": ms / steps, "launches_per_step": launches / steps, "share_of_kernel_time": ms / total}
        if name in traffic:
            gbs = traffic[name] * launches / (ms * 1e-3) / 1e9
            row.update({"pmc_bytes_per_step": traffic[name] * launches / steps, "pmc_GBps": gbs, "pmc_frac": gbs / HBM_PEAK_GBS})
        rows.append(row)
    return rows


def roofline_of(prof, info, n, n_docs, steps, traffic):
    """(dominant-kernel roofline, per-kernel list, kernel ms per step) from a profile report."""
    total_kernel_ms = sum(ms for _, ms in prof.values())
    dom_name, (dom_launches, dom_ms) = max(prof.items(), key=lambda kv: kv[1][1])
    bytes_per_step = kernel_bytes(dom_name, info, n, n_docs)
    roofline = {"kernel": dom_name, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "launches_per_step": dom_launches // steps, "avg_launch_ms": dom_ms / dom_launches,
                "share_of_kernel_time": dom_ms / total_kernel_ms, "traffic": traffic.get(dom_name)}
    if bytes_per_step is not None:
        achieved = bytes_per_step * steps / (dom_ms * 1e-3) / 1e9
        roofline.update({"achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": bytes_per_step / max(dom_launches // steps, 1)})
    else:
        roofline.update({"achieved": None, "frac": None})
    by_kernel = []
    for name, (launches, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:10]:
        b = kernel_bytes(name, info, n, n_docs)
        entry = {"kernel": name, "ms_per_step": ms / steps, "share_of_kernel_time": ms / total_kernel_ms}
        if b is not None:
            gbs = b * steps / (ms * 1e-3) / 1e9
            entry.update({"achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "traffic": traffic.get(name)})
        by_kernel.append(entry)
    per_step = {k: round(v[1] / steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
    # the kernel with the next largest share of the step, as a launch: two kernels may lie within a per cent of each other
    # (the 64 MiB document: the two pair passes of the radix sort together against the one launch of the fused finish), and
    # which of them leads can change from run to run -- the line carries both
    if len(by_kernel) > 1 and by_kernel[1].get("frac") is not None:
        name2 = by_kernel[1]["kernel"]
        launches2, ms2 = prof[name2]
        roofline["next"] = {"kernel": name2, "launches_per_step": launches2 // steps, "avg_launch_ms": ms2 / launches2,
                            "share_of_kernel_time": ms2 / total_kernel_ms, "achieved": by_kernel[1]["achieved"],
                            "frac": by_kernel[1]["frac"], "traffic": traffic.get(name2)}
        if ms2 > 0.9 * dom_ms:
            roofline["note"] = ("%s (%d launch(es) per step) and %s (%d) lie within 10 %% of each other in time per step: which of "
                                "them `roofline` names can change from run to run; `next` is the other one"
                                % (dom_name, dom_launches // steps, name2, launches2 // steps))
    return roofline, by_kernel, per_step


def hbm_fraction(prof, traffic):
    """The whole build: sum of (PMC bytes per launch x launches) / sum of HIP-event time over the build's kernels
    (those profiles/traffic.json has a figure for), as a fraction of the 8 TB/s peak."""
    num = den = 0.0
    covered = []
    for name, (launches, ms) in prof.items():
        if not name.startswith(BUILD_KERNEL_PREFIXES) or name not in traffic:
            continue
        num += traffic[name] * launches
        den += ms * 1e-3
        covered.append(name)
    if den == 0:
        return None
    build_ms = sum(ms for name, (_, ms) in prof.items() if name.startswith(BUILD_KERNEL_PREFIXES))
    return {"GBps": num / den / 1e9, "frac": num / den / 1e9 / HBM_PEAK_GBS, "kernels": len(covered),
            "share_of_build_kernel_time": den * 1e3 / build_ms if build_ms else None}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from east import hip_backend, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # EAST_BENCH_FORCE_DIST=1 runs the collective path even with one rank (used to smoke-test it on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("EAST_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # the shape: configs[1] on one GPU, the per-GPU shape of configs[3] on several
    scaling_shape = world > 1
    D = args.docs if args.docs is not None else (256 if scaling_shape else 1)
    doc_mib = args.doc_mib if args.doc_mib is not None else (1.0 if scaling_shape else 64.0)
    K = args.keyphrases if args.keyphrases is not None else (10000 if scaling_shape else 1000)

    # ---- synthetic inputs (seed = 20240 + config# + rank), outside the timed region ----
    doc_bytes = int(doc_mib * (1 << 20))
    n_bytes = doc_bytes * D                               # input bytes per GPU
    rng = np.random.default_rng(20240 + 2 + 1000 * rank)
    symbols, doc_offsets, n_strings, is_zipf = make_corpus(args, synthetic, rng, D, doc_bytes)
    m = int(n_strings.sum())
    n = int(symbols.size)
    # keyphrases: every rank contributes K/world sampled from its own documents, the set is replicated
    share = [K // world + (1 if r < K % world else 0) for r in range(world)]
    q_local = synthetic.keyphrases(rng, symbols, share[rank])
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, (q_local[0], q_local[1]))
    else:
        gathered = [q_local]
    q_parts, q_off = [], [0]
    for qs, qo in gathered:
        q_parts.append(qs)
        q_off.extend((qo[1:] + q_off[-1]).tolist())
    q_symbols, q_offsets = np.concatenate(q_parts), np.array(q_off, dtype=np.int64)

    d_symbols = torch.from_numpy(symbols.view(np.int32)).to(dev)          # resident in HBM
    local_block = torch.empty((K, D), dtype=torch.float64, device=dev)    # K x D_local
    full_table = torch.empty((world * K, D), dtype=torch.float64, device=dev) if use_dist else None

    index = hip_backend.HipIndex(local_rank, reserve_symbols=n)
    index.build_device(d_symbols.data_ptr(), n, doc_offsets, n_strings)    # also sizes the score scratch
    index.set_keyphrases(q_symbols, q_offsets)

    local_s = [0.0]                                       # host time of the rank-local part of the steps (build + score)
    gather_events = []                                    # (start, end) events around every all-gather

    def step():
        t_a = time.perf_counter()
        index.build_device(d_symbols.data_ptr(), n, doc_offsets, n_strings)
        index.score_resident(not args.denormalized, local_block.data_ptr())     # (returns when the block is written)
        local_s[0] += time.perf_counter() - t_a
        if use_dist:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_gather_into_tensor(full_table, local_block)           # RCCL over xGMI
            e1.record()
            gather_events.append((e0, e1))

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # Untimed profiling pass: EVERY kernel bracketed by HIP events on the library's stream -- the per-kernel
    # breakdown and the name of the dominant kernel.  (Two events per launch cost the stream time, about 10 % of
    # a 2 ms step with ~30 launches: that does not belong into the timed region.)
    profile_steps = max(1, min(args.steps, 3))
    index.profile_enable(True)
    fence()
    for _ in range(profile_steps):
        step()
    fence()
    prof = index.profile_report()
    index.profile_enable(False)
    dom_name = max(prof.items(), key=lambda kv: kv[1][1])[0]
    # Timed region: exactly args.steps steps; only the dominant kernel's launches are bracketed (its average
    # launch duration, live, over the timed region -- `roofline`).
    index.profile_only(dom_name)
    index.profile_enable(True)
    build_ms, score_ms = [], []
    fence()
    local_s[0] = 0.0
    del gather_events[:]
    step_wall_ms = []                                     # (a step ends synchronised: build read-back, score call)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t_s = time.perf_counter()
        step()
        step_wall_ms.append((time.perf_counter() - t_s) * 1e3)
        build_ms.append(index.last_build_ms)
        score_ms.append(index.last_score_ms)
    fence()
    elapsed = time.perf_counter() - t0
    prof_timed = index.profile_report()
    index.profile_enable(False)
    index.profile_only(None)
    step_local_ms = local_s[0] * 1e3 / args.steps
    allgather_ms = sum(a.elapsed_time(b) for a, b in gather_events) / args.steps if gather_events else 0.0
    rccl_world = None
    if use_dist:
        t = torch.tensor([elapsed, step_local_ms, allgather_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, step_local_ms, allgather_ms = (float(x) for x in t.tolist())
        rccl_world = dist.get_world_size()
    info = index.info()

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * n_bytes / (elapsed / args.steps)
        default_shape = (D, doc_mib, K, args.mode, args.corpus) == (1, 64.0, 1000, "text", "words") and not args.duplicate_docs
        zipf_shape = (D, doc_mib, K, args.mode, args.corpus) == (100, 1.0, 1000, "text", "zipf") and not args.duplicate_docs
        config2_shape = (D, doc_mib, K, args.mode, args.corpus) == (256, 1.0, 10000, "text", "words") and not args.duplicate_docs
        # (the counters were collected on these workloads: tools/profile_round.sh, profile_zipf.sh, profile_config2.sh)
        traffic_file = ("traffic.json" if default_shape else "traffic_zipf.json" if zipf_shape else
                        "traffic_config2.json" if config2_shape else None)
        traffic_meta = {}
        traffic = load_traffic(traffic_file, traffic_meta) if traffic_file else {}
        roofline, by_kernel, per_step = roofline_of(prof, info, n, D, profile_steps, traffic)
        live = prof_timed.get(dom_name)
        if live and live[0]:                             # the dominant kernel as measured inside the timed region
            roofline["launches_per_step"] = live[0] // args.steps
            roofline["avg_launch_ms"] = live[1] / live[0]
            b = kernel_bytes(dom_name, info, n, D)
            if b is not None:
                roofline["achieved"] = b * args.steps / (live[1] * 1e-3) / 1e9
                roofline["frac"] = roofline["achieved"] / HBM_PEAK_GBS
            roofline["avg_launch_ms_profiling_pass"] = prof[dom_name][1] / prof[dom_name][0]
        roofline["rocprof_hbm_fraction"] = hbm_fraction(prof, traffic)
        roofline["traffic_commit"] = traffic_meta.get("commit")
        roofline["traffic_source"] = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this "
                                      "workload (tools/profile_*.sh + summarize_profiles.py), committed with the code -- not "
                                      "measured by this run" % traffic_file) if traffic_file else None
        out = {
            "metric": "corpus chars/sec (SA+annotation build + keyphrase score table)",
            "value": value, "unit": "chars/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (symbols/indices) + f64 (scores)", "data": "synthetic",
            "config": {"workload": "%d synthetic %g MiB %s doc(s) per GPU (%s mode), "
                                   "%d keyphrases, %s scores, easa-HIP%s"
                                   % (D, doc_mib, "random-ASCII word-stream" if not is_zipf else
                                      "Zipf natural-language-like", args.mode, K,
                                      "denormalized" if args.denormalized else "normalized",
                                      " [BASELINE configs[3] per-GPU shape]" if scaling_shape else ""),
                       "symbols_per_gpu": n, "strings_per_gpu": m, "parallelism": "doc-shard x%d" % world,
                       "all_gather_bytes_per_rank": K * D * 8 if use_dist else 0},
            "step_ms_min": float(np.min(step_wall_ms)), "step_ms_median": float(np.median(step_wall_ms)),
            "step_ms_max": float(np.max(step_wall_ms)),
            "build_ms": float(np.mean(build_ms)), "score_ms": float(np.mean(score_ms)),
            "build_symbols_per_s": n / (float(np.mean(build_ms)) * 1e-3),
            "build_chars_per_s": n_bytes / (float(np.mean(build_ms)) * 1e-3),
            "keyphrase_scores_per_s": K * D / (float(np.mean(score_ms)) * 1e-3),
            "build_algorithmic_GBps": 16.0 * n / (float(np.mean(build_ms)) * 1e-3) / 1e9,
            "dc3_levels": info["dc3_levels"], "dc3_levels_resolved": info["dc3_levels_resolved"],
            "dc3_refine_rounds": info["refine_rounds"], "window_sorted": info["window_sorted"],
            "radix_passes": info["radix_passes"], "variable_length_keys": info.get("ht_keys", 0), "segmented_sort": info.get("seg_sort", 0),
            "roofline": roofline, "roofline_by_kernel": by_kernel, "kernels_ms_per_step": per_step,
            "kernel_launches_per_step": {k: v[0] // profile_steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
            "profiling_pass": "%d untimed step(s) with every kernel bracketed: roofline_by_kernel, kernels_ms_per_step, "
                              "rocprof_hbm_fraction; the timed steps bracket the dominant kernel only" % profile_steps,
            "lds_sorted": info.get("lds_sorted", 0),
        }
        if use_dist:
            # Multi-GPU accounting (the driver computes scaling efficiency itself from `value` at N = 1, 2, 4, 8 -- the
            # same-shape N = 1 base of this line is the `config2` leg of the N = 1 line, see below): the rank-local
            # part of a step (build + score, no collective; max over ranks), the all-gather behind it, and how much of
            # the step the local part is -- the figure a perfectly scaling run keeps at 1.
            out["multi_gpu"] = {"step_local_ms": step_local_ms, "allgather_ms": allgather_ms,
                                "local_fraction_of_step": step_local_ms / ms_per_step if ms_per_step else None,
                                "rccl_world_size": rccl_world, "backend": dist.get_backend(),
                                "scaling_efficiency": value / (world * args.base_value) if args.base_value > 0 else None,
                                "scaling_base_value": args.base_value if args.base_value > 0 else None,
                                "same_shape_single_gpu_base": "config2.value of the N=1 line (256 x 1 MiB docs, 10000 keyphrases)",
                                "note": "max over ranks; local_fraction_of_step = local / whole step (1 = the collective is free) -- NOT a "
                                        "scaling efficiency: that is value(N) / (N x the same-shape N=1 value), which the "
                                        "driver computes from the per-N lines"}
        else:
            out["scaling_base_note"] = ("--gpus N > 1 runs BASELINE configs[3]'s per-GPU shape (256 x 1 MiB documents, 10000 "
                                        "keyphrases per rank): its single-GPU base is this line's config2.value, not value")
        if world == 1 and not args.no_extras:
            out.update(first_build_leg(hip_backend, torch, local_rank, d_symbols, n, doc_offsets, n_strings, q_symbols,
                                       q_offsets, local_block, not args.denormalized, n_bytes))
        if world == 1:
            out["value_note"] = ("value = steady state: the timed steps rebuild the same collection on one handle (guesses "
                                 "from the build before always hold); value_first_build = a fresh handle's first build")
        if not args.no_extras:
            out.update(extras(index, prof, args, symbols, doc_offsets, n_strings, q_offsets, K, D, d_symbols))
            if "child_tables_ms" in out:
                # the reference's constructor also fills the child tables (easa.py:16-24); here they are built on first request
                out["build_ms_constructor"] = out["build_ms"] + out["child_tables_ms"]
        if "build_from_host_ms" in out:
            # "from symbols resident on host ... H2D included" (the reference's constructor starts from a
            # Python string, easa.py:16-24): east_hip_build from pageable host memory + the score call
            host_step_ms = out["build_from_host_ms"]["wall_ms_median"] + out["score_ms"]
            out["value_from_host"] = n_bytes / (host_step_ms * 1e-3)
            out["value_from_host_note"] = ("input bytes / (east_hip_build wall -- the median of twelve calls, the upload from pageable "
                                           "memory included -- + score call); value itself starts from symbols resident in HBM")
        if world == 1 and not args.no_config2 and default_shape:
            out["from_text"] = from_text_leg(hip_backend, synthetic, local_rank)
            ft = out["from_text"].get("ascii_64MiB")
            if ft:
                # raw text (Python bytes) -> prepared + indexed on the device (main.py:67-89 -> utils.py:31-79 ->
                # easa.py:16-24), + the score call: what `east keyphrases table` runs on this document
                out["value_from_text"] = ft["bytes"] / ((ft["wall_ms"] + out["score_ms"]) * 1e-3)
            out["worst_case"] = worst_case_leg(hip_backend, synthetic, torch, dev, local_rank, not args.no_cpu_baseline)
            out["config2"] = config2_leg(args, hip_backend, synthetic, torch, dev, local_rank)
            out["config5"] = config5_leg(args, hip_backend, synthetic, torch, dev, local_rank)
            out["config5_prose"] = config5_prose_leg(hip_backend, synthetic, torch, dev, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, synthetic)
    if use_dist and not args.no_in_process:
        # The path `east -g N` takes by default is not the one timed above (one process per GPU, torch.distributed) but the
        # library's in-process group: one host thread per device, ncclCommInitAll, one grouped ncclAllGather (csrc/multi.h).
        # Every rank gives its device memory back and waits on a CPU-side (gloo) barrier -- an NCCL barrier would keep a
        # kernel spinning on the very devices rank 0 is about to use --, rank 0 runs the same shard shapes through
        # east_hip_group_build + east_hip_score_table_multi on devices 0 .. N-1, then everybody goes on.
        # The leg runs in a CHILD process of rank 0 under a time limit (this code path has never seen two GPUs: a
        # communicator that hangs must not take the torch leg's numbers with it), and the waits around it are bounded.
        import datetime
        index.close()
        del d_symbols, local_block, full_table
        torch.cuda.empty_cache()
        leg = None
        try:
            ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=900))
            dist.monitored_barrier(group=ctl, timeout=datetime.timedelta(seconds=300))
            if rank == 0:
                leg = in_process_child(args, world, D, doc_mib, K)
            dist.monitored_barrier(group=ctl, timeout=datetime.timedelta(seconds=900))
        except Exception as e:                             # noqa: BLE001 (the line must still go out: the torch leg stands)
            if rank == 0 and leg is None:
                leg = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        if rank == 0:
            out.setdefault("multi_gpu", {})["in_process"] = leg
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line goes last: native libraries (the RCCL version banner) write through C stdio, whose
        # buffer would otherwise be flushed behind it when stdout is a pipe
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        detail_path = write_detail(out)
        print(json.dumps(out) if args.full_line else compact_line(out, detail_path), flush=True)


LINE_LIMIT = 6144          # the driver keeps an 8 KB tail of stdout and parses its LAST line: that line stays under 6 KB


def _r(x, digits=5):
    """Numbers of the line: `digits` significant digits (the full figures are in bench_detail.json)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _leg_summary(leg):
    """One line per secondary leg: value, step, build, score, the dominant kernel and its fraction of the HBM peak."""
    if not isinstance(leg, dict):
        return None
    res = _pick(leg, ("value", "ms_per_step", "build_ms", "score_ms", "first_build_ms", "refine_rounds"))
    roof = leg.get("roofline")
    rows = leg.get("roofline_by_kernel")
    if isinstance(roof, dict):
        res["kernel"], res["frac"] = roof.get("kernel"), roof.get("frac")
    elif rows:                                           # (the config 5 legs: counter fractions per kernel)
        res["kernel"], res["frac"] = rows[0].get("kernel"), rows[0].get("pmc_frac")
    elif isinstance(leg.get("kernels_ms_per_step"), dict) and leg["kernels_ms_per_step"]:
        res["kernel"] = next(iter(leg["kernels_ms_per_step"]))
    whole = leg.get("rocprof_hbm_fraction") or (roof or {}).get("rocprof_hbm_fraction")
    if isinstance(whole, dict):
        res["build_hbm_frac"] = whole.get("frac")
    if isinstance(leg.get("roofline_score"), dict):
        res["score_frac"] = leg["roofline_score"].get("frac")
    if isinstance(leg.get("from_text"), dict):
        res["from_text_ms"] = leg["from_text"].get("wall_ms")
    return res


def compact_line(out, detail_path=None):
    """The ONE line the driver parses (its stdout tail is 8 KB; the round-5 line had grown to 21.7 KB and was not
    parsed): the contract keys, `roofline`, `cpu_baseline`, the host-resident / first-build / raw-text values and a
    one-line summary per secondary leg.  Everything else -- per-kernel tables, notes, every repetition -- stays in
    `bench_detail.json` (write_detail).  Pure function of the assembled dict: tests/test_host_logic.py runs it on canned
    numbers and bounds its length."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    line["vs_baseline"] = out.get("vs_baseline")
    line.update(_pick(out, ("dtype", "data", "config", "step_ms_min", "step_ms_median", "step_ms_max", "build_ms", "score_ms",
                            "keyphrase_scores_per_s")))
    roof = out.get("roofline") or {}
    r = _pick(roof, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "avg_launch_ms",
                     "algorithmic_bytes_per_launch", "share_of_kernel_time", "traffic_commit"))
    if "traffic" not in r:
        r["traffic"] = None
    if isinstance(roof.get("next"), dict):
        r["next"] = _pick(roof["next"], ("kernel", "launches_per_step", "avg_launch_ms", "achieved", "frac", "traffic"))
    if isinstance(roof.get("rocprof_hbm_fraction"), dict):
        r["rocprof_hbm_fraction"] = _pick(roof["rocprof_hbm_fraction"], ("GBps", "frac", "kernels"))
    line["roofline"] = r
    if isinstance(out.get("roofline_score"), dict):
        line["roofline_score"] = _pick(out["roofline_score"], ("kernel", "avg_launch_ms", "algorithmic_bytes_per_launch",
                                                               "achieved", "frac"))
    cpu = out.get("cpu_baseline")
    if isinstance(cpu, dict):
        c = _pick(cpu, ("value", "unit", "cores", "kind", "sample", "keyphrase_scores_per_s"))
        if isinstance(cpu.get("over_documents"), dict):
            c["over_documents"] = _pick(cpu["over_documents"], ("value", "cores"))
        line["cpu_baseline"] = c
    line.update(_pick(out, ("value_from_host", "value_first_build", "value_from_text", "first_build_ms",
                            "build_ms_without_guesses", "child_tables_ms", "build_ms_constructor")))
    host = out.get("build_from_host_ms")
    if isinstance(host, dict):
        line["build_from_host_ms"] = _pick(host, ("wall_ms_median", "wall_ms_min", "wall_ms_max", "h2d_bytes", "bytes_per_symbol"))
    ft = (out.get("from_text") or {})
    if ft:
        line["from_text"] = {name: _pick(leg, ("wall_ms", "prep_ms", "build_ms", "chars_per_s")) for name, leg in ft.items()}
    for name in ("config2", "config5", "config5_prose"):
        if name in out:
            line[name] = _leg_summary(out[name])
    wc = out.get("worst_case")
    if isinstance(wc, dict):
        line["worst_case"] = {
            "cases": [_pick(c, ("n", "symbols", "build_ms", "chars_per_s", "random_text_same_size_build_ms",
                                "speedup_vs_oracle_one_core")) for c in wc.get("cases", [])],
            "long_repeats_16Mi_ms": {c["input"]: c["build_ms"] for c in wc.get("long_repeats", [])}}
    mg = out.get("multi_gpu")
    if isinstance(mg, dict):
        line["multi_gpu"] = _pick(mg, ("step_local_ms", "allgather_ms", "local_fraction_of_step", "rccl_world_size", "backend",
                                       "scaling_efficiency", "scaling_base_value", "in_process"))
    if detail_path:
        line["detail"] = os.path.basename(detail_path)
    line = _r(line)
    text = json.dumps(line, separators=(",", ":"))
    # (never reached with the keys above -- about 4 KB --; a future key must not push the line past the driver's tail)
    for victim in ("worst_case", "from_text", "config5_prose", "config5", "config2", "roofline_score"):
        if len(text) <= LINE_LIMIT:
            break
        line.pop(victim, None)
        line["dropped_for_length"] = line.get("dropped_for_length", []) + [victim]
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:
        raise RuntimeError("bench line of %d bytes: the driver reads an 8 KB tail" % len(text))
    return text


def write_detail(out):
    """Everything the run measured, as one JSON document next to bench.py (and under gpurun_out/ when that directory
    exists, so that it comes back from the GPU box).  Returns the path written, or None."""
    written = None
    paths = [os.path.join(ROOT, "bench_detail.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    if os.environ.get("EAST_BENCH_DETAIL"):                 # (the profiling scripts keep one per run: tools/profile_*.sh)
        paths.append(os.environ["EAST_BENCH_DETAIL"])
    for path in paths:
        try:
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
            written = written or path
        except OSError:
            pass
    return written


def in_process_child(args, world, D, doc_mib, K, limit_s=600):
    """Rank 0: the in-process leg in a child process (fresh interpreter, no torch, no process group), under a time limit.
    Returns the leg's record or {"error": ...}."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--in-process-child", str(world), "--docs", str(D), "--doc-mib", repr(doc_mib),
           "--keyphrases", str(K), "--mode", args.mode, "--corpus", args.corpus] + (["--denormalized"] if args.denormalized else [])
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                        "TORCHELASTIC_RUN_ID", "EAST_HIP_DEVICE")}      # (device masks of the whole job, if any, stay)
    try:
        done = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit_s)
    except subprocess.TimeoutExpired:
        return {"error": "the in-process leg did not finish within %d s" % limit_s}
    lines = [ln for ln in done.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if done.returncode != 0 or not lines:
        return {"error": "child exit code %d: %s" % (done.returncode, done.stderr.decode(errors="replace")[-300:])}
    return json.loads(lines[-1])


def in_process_child_main(argv):
    """`bench.py --in-process-child N --docs D --doc-mib M --keyphrases K ...`: what in_process_child() starts.  Makes rank
    0's shard again (same seeds as main()), runs in_process_leg, prints its record as one JSON line."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--in-process-child", type=int, required=True)
    ap.add_argument("--docs", type=int, required=True)
    ap.add_argument("--doc-mib", type=float, required=True)
    ap.add_argument("--keyphrases", type=int, required=True)
    ap.add_argument("--mode", default="text")
    ap.add_argument("--corpus", default="words")
    ap.add_argument("--duplicate-docs", type=int, default=0)
    ap.add_argument("--denormalized", action="store_true")
    args = ap.parse_args(argv)
    from east import hip_backend, synthetic
    world, D, K = args.in_process_child, args.docs, args.keyphrases
    doc_bytes = int(args.doc_mib * (1 << 20))
    rng = np.random.default_rng(20240 + 2)                  # (rank 0's seed)
    symbols, doc_offsets, n_strings, _ = make_corpus(args, synthetic, rng, D, doc_bytes)
    q_symbols, q_offsets = synthetic.keyphrases(rng, symbols, K)
    leg = in_process_leg(hip_backend, world, symbols, doc_offsets, n_strings, q_symbols, q_offsets, not args.denormalized, doc_bytes * D)
    sys.stdout.flush()
    print(json.dumps(leg), flush=True)


def in_process_leg(hip_backend, world, symbols, doc_offsets, n_strings, q_symbols, q_offsets, normalized, n_bytes, steps=3):
    """The in-process device group (east_hip_group_build + east_hip_score_table_multi: what `east -g N` runs by default)
    on devices 0 .. world-1, on `world` copies of rank 0's shard -- the same per-device shapes as the torch leg.  The
    group's entry points take HOST symbols and return the K x D table on the host: a step here includes the uploads
    (bytes or 16-bit words through each shard's pinned ring) and the copy of the table, which the torch leg's
    device-resident steps do not."""
    sym_all = np.tile(symbols, world)
    n = int(symbols.size)
    off_all = np.concatenate([doc_offsets[:-1] + r * n for r in range(world)] + [[world * n]]).astype(np.int64)
    ms_all = np.tile(n_strings, world)
    group = hip_backend.HipGroup(list(range(world)))
    try:
        group.build(sym_all, off_all, ms_all)              # (arena allocation, the rings' pinning: not timed)
        table = group.score_table(q_symbols, q_offsets, normalized)
        time.sleep(0.1)                                    # (the rings are pinned in the background)
        walls, parts = [], []
        for _ in range(steps):
            t0 = time.perf_counter()
            group.build(sym_all, off_all, ms_all)
            table = group.score_table(q_symbols, q_offsets, normalized)
            walls.append((time.perf_counter() - t0) * 1e3)
            parts.append(group.info())
        best = int(np.argmin(walls))
        shards = np.diff(group.first_doc).tolist()
        return {"step_ms": float(np.median(walls)), "step_ms_min": float(min(walls)), "build_ms": parts[best]["build_ms"],
                "score_ms": parts[best]["score_ms"], "gather_ms": parts[best]["gather_ms"], "gather": parts[best]["gather"],
                "rccl_ranks": world if parts[best]["gather"] == "rccl" else 0, "shards": shards,
                "value": world * n_bytes / (float(np.median(walls)) * 1e-3), "from_host": True, "steps": steps,
                "table_shape": list(table.shape)}
    finally:
        group.close()


def extras(index, prof, args, symbols, doc_offsets, n_strings, q_offsets, K, D, d_symbols=None):
    """Untimed extras of the same index: score roofline from a counted run, the child-table kernel, the build
    from host-resident symbols."""
    res = {}
    n_q = int(q_offsets[-1])
    walk = prof.get("score_walk_kernel")
    if walk:
        probes = index.score_probes(not args.denormalized)
        # 8 B per table read / probe (suffix-array entry + symbol), 8 B per score; the per-suffix results written by the
        # walk and read by the reduction (8 B each way) only where the reduction kernel still runs
        b = 8 * probes + (16 * n_q * D if "score_reduce_kernel" in prof else 0) + 8 * K * D
        ms = walk[1] / walk[0]
        res["roofline_score"] = {"kernel": "score_walk_kernel", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "probes": probes, "algorithmic_bytes_per_launch": b, "avg_launch_ms": ms,
                                 "achieved": b / (ms * 1e-3) / 1e9, "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "note": "random 4-byte reads in 64-byte sectors: latency-bound, not bandwidth-bound"}
    # child tables: built by the first east_hip_get_tables request that asks for them
    index.profile_enable(True)
    index.tables(0, names=("childtab_up",))
    rep = index.profile_report()
    index.profile_enable(False)
    child = [rep[k] for k in ("child_stream_kernel", "child_wide_kernel") if k in rep]
    if child:
        res["child_tables_ms"] = sum(ms / launches for launches, ms in child)
    # the build as a handle's FIRST build runs it: nothing guessed from a build before, every read-back in place
    # (the timed steps rebuild the same collection, so their guesses -- alphabet, no large tie groups -- always hold)
    if d_symbols is not None:
        lib = index._lib
        lib.east_hip_debug_set_speculation(0)
        times = []
        for _ in range(3):
            index.build_device(d_symbols.data_ptr(), int(symbols.size), doc_offsets, n_strings)
            times.append(index.last_build_ms)
        lib.east_hip_debug_set_speculation(1)
        res["build_ms_without_guesses"] = min(times)
    # the build from host-resident symbols (east_hip_build: one 4 B/symbol H2D copy in front), wall clock
    # (twelve calls: the first one pins the upload ring)
    walls = []
    for _ in range(12):
        t0 = time.perf_counter()
        index.build(symbols, doc_offsets, n_strings)
        walls.append((time.perf_counter() - t0) * 1e3)
    narrow = int(index.info().get("narrow_upload", 0))
    per_symbol = {0: 4, 1: 2, 2: 1}[narrow]
    res["build_from_host_ms"] = {"wall_ms_min": min(walls), "wall_ms_median": sorted(walls)[len(walls) // 2], "wall_ms_first_call": walls[0],
                                 "wall_ms_max": max(walls[2:]), "wall_ms_calls": [round(w, 2) for w in walls],
                                 "device_build_ms": index.last_build_ms, "h2d_bytes": int(symbols.size) * per_symbol,
                                 "bytes_per_symbol": per_symbol,
                                 "note": "pageable host memory in, index out.  Host threads narrow the symbols -- to bytes while all "
                                         "text lies below 0xFF, else to 16-bit words --, send them through a pinned ring (pinned by the "
                                         "handle's first such call, in line) and a kernel widens them on the device (east_hip.hip: "
                                         "upload_symbols_narrow; the link: 56 GB/s, tools/pcie_probe.py); wall_ms_max: the slowest call "
                                         "after the first two"}
    return res


def first_build_leg(hip_backend, torch, local_rank, d_symbols, n, doc_offsets, n_strings, q_symbols, q_offsets, block,
                    normalized, n_bytes, reps=3):
    """A handle's FIRST build and score: a fresh handle every time, nothing known from a build before -- the window
    width and the fused finish planned from a sample of the text itself, every read-back in place.  This is what one
    `east keyphrases table` process runs; the timed steps of the main line rebuild the same collection on one handle
    (steady state: their guesses always hold).  The arena is allocated when the handle is created, outside the timing."""
    b_dev, b_wall, s_wall = [], [], []
    info = {}
    for r in range(reps + 1):                               # (the first repetition pays first-touch costs: dropped)
        index = hip_backend.HipIndex(local_rank, reserve_symbols=n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        index.build_device(d_symbols.data_ptr(), n, doc_offsets, n_strings)
        t1 = time.perf_counter()
        index.set_keyphrases(q_symbols, q_offsets)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        index.score_resident(normalized, block.data_ptr())
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if r > 0:
            b_dev.append(index.last_build_ms)
            b_wall.append((t1 - t0) * 1e3)
            s_wall.append((t3 - t2) * 1e3)
        info = index.info()
        index.close()
    step = float(np.mean(b_wall)) + float(np.mean(s_wall))
    return {"first_build_ms": float(np.mean(b_dev)), "first_build_wall_ms": float(np.mean(b_wall)),
            "first_score_wall_ms": float(np.mean(s_wall)), "value_first_build": n_bytes / (step * 1e-3),
            "first_build_plan": {"fused_finish": info.get("fused_finish"), "u64_passes": info.get("radix_passes_u64"),
                                 "u32_passes": info.get("radix_passes_u32"), "refine_rounds": info.get("refine_rounds"),
                                 "variable_length_keys": info.get("ht_keys", 0), "segmented_sort": info.get("seg_sort", 0)},
            "first_build_note": "fresh handle per repetition (%d), no hints from earlier builds; value_first_build = input "
                                "bytes / (build wall + score wall)" % reps}


def from_text_leg(hip_backend, synthetic, local_rank):
    """What `east keyphrases table` runs before it can score (reference east/main.py:67-89 -> utils.py:31-79): raw
    text as Python `bytes` -> prepared, indexed collection on the device, through east_hip_build_texts[_v] (upload of
    the raw bytes, text preparation kernels, build).  Wall clock around HipIndex.build_texts; configs[1]'s 64 MiB
    ASCII word stream, and 64 documents of 1 MiB drawn from the prose that ships with the image (if there is any)."""
    res = {}
    hip_backend.unicode_tables()
    cases = []
    text = synthetic.word_stream_document(np.random.default_rng(20240 + 2), 64 << 20)[0]
    cases.append(("ascii_64MiB", [text]))
    raw, _ = synthetic.image_prose(24 << 20, False)
    if len(raw) > (2 << 20):
        rng = np.random.default_rng(20245)
        lines = [ln for ln in raw.split(b"\n") if len(ln) > 20]
        lens = np.array([len(ln) + 1 for ln in lines])
        picks = rng.integers(0, len(lines), size=int((64 << 20) / lens.mean()) + 1)
        big = b"\n".join(lines[i] for i in picks)[:64 << 20]
        cases.append(("prose_64x1MiB", [big[i:i + (1 << 20)] for i in range(0, len(big), 1 << 20)]))
    for name, texts in cases:
        n_bytes = sum(len(t) for t in texts)
        firsts, walls, preps, builds = [], [], [], []
        for rep in range(3):                                # fresh handles: the first call of a process (+ arena allocation)
            index = hip_backend.HipIndex(local_rank)
            t0 = time.perf_counter()
            index.build_texts(texts)
            firsts.append((time.perf_counter() - t0) * 1e3)
            if rep == 2:
                for _ in range(3):                          # the same handle again: allocations in place
                    t0 = time.perf_counter()
                    index.build_texts(texts)
                    walls.append((time.perf_counter() - t0) * 1e3)
                    preps.append(index.last_prep_ms)
                    builds.append(index.last_build_ms)
                info = index.info()
                # what the preparation consists of: one more call with every kernel bracketed (untimed)
                index.profile_enable(True)
                index.build_texts(texts)
                prof = index.profile_report()
                index.profile_enable(False)
                prep_kernels = {k: round(v[1], 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])
                                if k.startswith(("tp_", "scan_"))}
            index.close()
        res[name] = {"bytes": n_bytes, "docs": len(texts), "symbols": info["n_total"],
                     "wall_ms": min(walls), "prep_ms": min(preps), "build_ms": min(builds),
                     "prep_kernels_ms": prep_kernels, "prep_kernels_total_ms": round(sum(prep_kernels.values()), 4),
                     "variable_length_keys": info.get("ht_keys", 0), "segmented_sort": info.get("seg_sort", 0), "refine_rounds": info["refine_rounds"],
                     "first_call_wall_ms": min(firsts[1:]), "chars_per_s": n_bytes / (min(walls) * 1e-3),
                     "note": "wall = Python bytes -> finished index (H2D of the raw text from pageable memory included); "
                             "first_call = fresh handle, arena allocation included"}
    return res


def worst_case_leg(hip_backend, synthetic, torch, dev, local_rank, with_oracle):
    """The reference's own (and only) benchmark workload: analysis/runtime.py:19-31 times AST.get_ast on
    analysis/utils.py:5-9 `worst_case_strings_collection(m = 100, n)` -- as shipped, 100 IDENTICAL strings of n - 4
    letters.  Every suffix sits in a tie group of 100 that only the terminators tell apart and the common prefixes are as
    long as the strings: the input that drives the tie-refinement rounds into prefix doubling, the capped LCP comparison
    into the blocked-Kasai finish, or the build into the DC3 fallback.  Per n: the build (a fresh handle's first build
    and the steady state), which path ran, the same number of symbols of random word-stream text beside it, and the
    oracle's time on the same input."""
    res = {"workload": "analysis/utils.py worst_case_strings_collection(m=100, n): 100 identical strings of n-4 letters, "
                       "one document (AST.get_ast), symbols resident in HBM", "cases": []}
    for n in (100, 1000, 10000, 100000):
        rng = np.random.default_rng(20240 + 6)
        sym, m = synthetic.worst_case_collection(rng, 100, n)
        off, ms = np.array([0, sym.size], dtype=np.int64), np.array([m], dtype=np.int32)
        d_sym = torch.from_numpy(sym.view(np.int32)).to(dev)
        index = hip_backend.HipIndex(local_rank, reserve_symbols=int(sym.size))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        index.build_device(d_sym.data_ptr(), sym.size, off, ms)
        first_wall = (time.perf_counter() - t0) * 1e3
        first_dev = index.last_build_ms
        times = []
        for _ in range(3):
            index.build_device(d_sym.data_ptr(), sym.size, off, ms)
            times.append(index.last_build_ms)
        info = index.info()
        index.close()
        # random word-stream text with the same number of symbols (text mode: 3-word strings), same code
        _, rsym, rm = synthetic.word_stream_document(np.random.default_rng(20240 + 7), int(sym.size * 1.07), want_text=False)
        d_r = torch.from_numpy(rsym.view(np.int32)).to(dev)
        rindex = hip_backend.HipIndex(local_rank, reserve_symbols=int(rsym.size))
        rtimes = []
        for _ in range(4):
            rindex.build_device(d_r.data_ptr(), rsym.size, np.array([0, rsym.size]), np.array([rm]))
            rtimes.append(rindex.last_build_ms)
        rindex.close()
        per_sym = min(times) / sym.size
        per_sym_random = min(rtimes[1:]) / rsym.size
        case = {"n": n, "m": 100, "symbols": int(sym.size), "build_ms": min(times), "first_build_ms": first_dev,
                "first_build_wall_ms": first_wall, "chars_per_s": int(sym.size - m) / (min(times) * 1e-3),
                "path": {"window_sorted": info["window_sorted"], "refine_rounds": info["refine_rounds"],
                         "dc3_levels": info["dc3_levels"], "radix_passes": info["radix_passes"],
                         "lds_sorted": info.get("lds_sorted", 0), "fused_finish": info.get("fused_finish", 0)},
                "random_text_same_size_build_ms": min(rtimes[1:]), "random_text_symbols": int(rsym.size),
                "ns_per_symbol": per_sym * 1e6, "ns_per_symbol_random_text": per_sym_random * 1e6,
                "slowdown_vs_random_text_per_symbol": per_sym / per_sym_random}
        if with_oracle:
            from oracle import easa_oracle
            t0 = time.perf_counter()
            easa_oracle.OracleEASA(symbols=sym, n_strings=m)
            case["oracle_build_s"] = time.perf_counter() - t0
            case["speedup_vs_oracle_one_core"] = case["oracle_build_s"] * 1e3 / min(times)
        res["cases"].append(case)
    # long repeats of other shapes, 16 Mi symbols in one string (get_ast([one string])): a run of one letter, a period of
    # three, a passage written 16 times, a Fibonacci string -- the inputs on which a step that restarts comparisons inside
    # repeats goes quadratic (the finishing pass of the LCP table did: 0.8 s)
    n = 16 << 20
    rng = np.random.default_rng(20240 + 8)
    fa, fb = np.array([65], np.uint32), np.array([65, 66], np.uint32)
    while fb.size < n:
        fa, fb = fb, np.concatenate([fb, fa])
    shapes = {"random A-Z": rng.integers(65, 91, size=n, dtype=np.uint32), "one letter": np.full(n, 65, np.uint32),
              "period of three": np.resize(np.array([65, 66, 67], np.uint32), n),
              "16 copies of a 1 MiB passage": np.tile(rng.integers(65, 91, size=n // 16, dtype=np.uint32), 16),
              "Fibonacci string": fb[:n]}
    index = hip_backend.HipIndex(local_rank, reserve_symbols=n + 1)
    res["long_repeats"] = []
    base_ms = None
    for name, body in shapes.items():
        sym = np.concatenate([body, [synthetic.TERMINATOR_START]]).astype(np.uint32)
        d_sym = torch.from_numpy(sym.view(np.int32)).to(dev)
        times = []
        for _ in range(3):
            index.build_device(d_sym.data_ptr(), sym.size, np.array([0, sym.size]), np.array([1]))
            times.append(index.last_build_ms)
        info = index.info()
        base_ms = base_ms or min(times)
        res["long_repeats"].append({"input": name, "symbols": int(sym.size), "build_ms": min(times), "ns_per_symbol": min(times) * 1e6 / sym.size,
                                    "slowdown_vs_random_text": min(times) / base_ms, "refine_rounds": info["refine_rounds"],
                                    "window_sorted": info["window_sorted"], "dc3_levels": info["dc3_levels"]})
    index.close()
    return res


def config5_leg(args, hip_backend, synthetic, torch, dev, local_rank):
    """BASELINE config 5 stand-in: 100 x 1 MiB natural-language-like documents (Zipf word stream), 1 000 keyphrases --
    the configuration in which the tie-refinement rounds carry the build."""
    D, K, steps = 100, 1000, 3
    rng = np.random.default_rng(20240 + 5)
    vocab = synthetic.zipf_vocabulary(np.random.default_rng(20245))
    parts, ms = zip(*[synthetic.zipf_document(rng, 1 << 20, vocab) for _ in range(D)])
    symbols = np.concatenate(parts)
    doc_offsets = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.int64)
    n_strings = np.array(ms, dtype=np.int32)
    n = int(symbols.size)
    qs, qo = synthetic.keyphrases(rng, symbols, K)
    d_symbols = torch.from_numpy(symbols.view(np.int32)).to(dev)
    block = torch.empty((K, D), dtype=torch.float64, device=dev)
    index = hip_backend.HipIndex(local_rank, reserve_symbols=n)
    index.build_device(d_symbols.data_ptr(), n, doc_offsets, n_strings)
    index.set_keyphrases(qs, qo)
    index.score_resident(True, block.data_ptr())
    build_ms, score_ms = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        index.build_device(d_symbols.data_ptr(), n, doc_offsets, n_strings)
        index.score_resident(True, block.data_ptr())
        build_ms.append(index.last_build_ms)
        score_ms.append(index.last_score_ms)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    index.profile_enable(True)                            # (the per-kernel breakdown: the same steps again, untimed)
    for _ in range(steps):
        index.build_device(d_symbols.data_ptr(), n, doc_offsets, n_strings)
        index.score_resident(True, block.data_ptr())
    torch.cuda.synchronize()
    prof = index.profile_report()
    index.profile_enable(False)
    info = index.info()
    per_step = {k: round(v[1] / steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
    index.close()
    first = first_build_leg(hip_backend, torch, local_rank, d_symbols, n, doc_offsets, n_strings, qs, qo, block, True,
                            D * (1 << 20), reps=2)
    traffic = load_traffic("traffic_zipf.json")             # PMC passes of this workload (tools/profile_zipf.sh)
    return {"first_build_ms": first["first_build_ms"], "value_first_build": first["value_first_build"],
            "first_build_plan": first["first_build_plan"],
            "roofline_by_kernel": pmc_by_kernel(prof, steps, traffic), "rocprof_hbm_fraction": hbm_fraction(prof, traffic),
            "workload": "100 synthetic 1 MiB Zipf natural-language-like docs (text mode), 1000 keyphrases, normalized",
            "steps": steps, "ms_per_step": elapsed * 1e3 / steps, "value": D * (1 << 20) / (elapsed / steps), "unit": "chars/s",
            "symbols": n, "build_ms": float(np.mean(build_ms)), "score_ms": float(np.mean(score_ms)),
            "refine_rounds": info["refine_rounds"], "lds_sorted": info.get("lds_sorted", 0),
            "window_sorted": info["window_sorted"], "kernels_ms_per_step": dict(list(per_step.items())[:12])}


def config5_prose_leg(hip_backend, synthetic, torch, dev, local_rank):
    """BASELINE config 5, second stand-in: 100 x 1 MiB of prose-like text from the order-3 character model of
    east/synthetic.py (trained on the image's prose: English letter statistics, frequent words and word pairs; the Zipf
    stand-in of `config5` has uniform letters), 1 000 keyphrases.  Two figures: raw text (Python bytes) -> finished index
    through the device text preparation, wall clock; and the steady-state step on the prepared symbols resident in HBM."""
    D, K, steps = 100, 1000, 3
    texts = synthetic.prose_like_texts(np.random.default_rng(20240 + 5), D, 1 << 20)
    index = hip_backend.HipIndex(local_rank)
    walls = []
    for _ in range(4):
        t0 = time.perf_counter()
        index.build_texts(texts)
        walls.append((time.perf_counter() - t0) * 1e3)
    prep_ms, text_build_ms = index.last_prep_ms, index.last_build_ms
    symbols, doc_offsets, n_strings = index.prepared()
    index.close()
    n = int(symbols.size)
    rng = np.random.default_rng(20240 + 5)
    qs, qo = synthetic.keyphrases(rng, symbols, K)
    d_symbols = torch.from_numpy(symbols.view(np.int32)).to(dev)
    block = torch.empty((K, D), dtype=torch.float64, device=dev)
    index = hip_backend.HipIndex(local_rank, reserve_symbols=n)
    index.build_device(d_symbols.data_ptr(), n, doc_offsets, n_strings)
    first_build_ms = index.last_build_ms
    index.set_keyphrases(qs, qo)
    index.score_resident(True, block.data_ptr())
    build_ms, score_ms = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        index.build_device(d_symbols.data_ptr(), n, doc_offsets, n_strings)
        index.score_resident(True, block.data_ptr())
        build_ms.append(index.last_build_ms)
        score_ms.append(index.last_score_ms)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    index.profile_enable(True)
    for _ in range(steps):
        index.build_device(d_symbols.data_ptr(), n, doc_offsets, n_strings)
        index.score_resident(True, block.data_ptr())
    torch.cuda.synchronize()
    prof = index.profile_report()
    index.profile_enable(False)
    info = index.info()
    index.close()
    per_step = {k: round(v[1] / steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
    return {"workload": "100 x 1 MiB prose-like text (order-3 character model trained on the image's prose), text mode, 1000 "
                        "keyphrases, normalized",
            "steps": steps, "ms_per_step": elapsed * 1e3 / steps, "value": D * (1 << 20) / (elapsed / steps), "unit": "chars/s",
            "symbols": n, "strings": int(np.sum(n_strings)), "build_ms": float(np.mean(build_ms)), "score_ms": float(np.mean(score_ms)),
            "first_build_ms": first_build_ms,
            "from_text": {"wall_ms": min(walls[1:]), "prep_ms": prep_ms, "build_ms": text_build_ms,
                          "chars_per_s": D * (1 << 20) / (min(walls[1:]) * 1e-3)},
            "refine_rounds": info["refine_rounds"], "variable_length_keys": info.get("ht_keys", 0), "segmented_sort": info.get("seg_sort", 0),
            "lds_sorted": info.get("lds_sorted", 0), "first_kept": info.get("first_kept"), "kernels_ms_per_step": dict(list(per_step.items())[:12])}


def config2_leg(args, hip_backend, synthetic, torch, dev, local_rank):
    """BASELINE configs[2]: 256 x 1 MiB documents, 10 000 keyphrases, one GPU -- the score-kernel configuration."""
    D, K, steps = 256, 10000, 3
    rng = np.random.default_rng(20240 + 3)
    symbols, doc_offsets, n_strings, _ = make_corpus(args, synthetic, rng, D, 1 << 20)
    n = int(symbols.size)
    qs, qo = synthetic.keyphrases(rng, symbols, K)
    d_symbols = torch.from_numpy(symbols.view(np.int32)).to(dev)
    block = torch.empty((K, D), dtype=torch.float64, device=dev)
    index = hip_backend.HipIndex(local_rank, reserve_symbols=n)
    index.build_device(d_symbols.data_ptr(), n, doc_offsets, n_strings)
    index.set_keyphrases(qs, qo)
    index.score_resident(True, block.data_ptr())
    build_ms, score_ms = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        index.build_device(d_symbols.data_ptr(), n, doc_offsets, n_strings)
        index.score_resident(True, block.data_ptr())
        build_ms.append(index.last_build_ms)
        score_ms.append(index.last_score_ms)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    index.profile_enable(True)                            # (the per-kernel breakdown: the same steps again, untimed)
    for _ in range(steps):
        index.build_device(d_symbols.data_ptr(), n, doc_offsets, n_strings)
        index.score_resident(True, block.data_ptr())
    torch.cuda.synchronize()
    prof = index.profile_report()
    index.profile_enable(False)
    info = index.info()
    traffic = load_traffic("traffic_config2.json")          # PMC passes of this workload (tools/profile_config2.sh)
    roofline, by_kernel, per_step = roofline_of(prof, info, n, D, steps, traffic)
    roofline["traffic_source"] = "profiles/traffic_config2.json (tools/profile_config2.sh), committed counters of an earlier run"
    res = {"workload": "256 synthetic 1 MiB random-ASCII word-stream docs (text mode), 10000 keyphrases, normalized",
           "steps": steps, "ms_per_step": elapsed * 1e3 / steps, "value": D * (1 << 20) / (elapsed / steps),
           "unit": "chars/s", "symbols": n, "build_ms": float(np.mean(build_ms)), "score_ms": float(np.mean(score_ms)),
           "build_chars_per_s": D * (1 << 20) / (float(np.mean(build_ms)) * 1e-3),
           "keyphrase_scores_per_s": K * D / (float(np.mean(score_ms)) * 1e-3),
           "roofline": roofline, "kernels_ms_per_step": dict(list(per_step.items())[:12])}
    walk = prof.get("score_walk_kernel")
    if walk and not args.no_extras:
        probes = index.score_probes(True)
        b = 8 * probes + (16 * int(qo[-1]) * D if "score_reduce_kernel" in prof else 0) + 8 * K * D
        ms = walk[1] / walk[0]
        res["roofline_score"] = {"kernel": "score_walk_kernel", "probes": probes, "algorithmic_bytes_per_launch": b,
                                 "avg_launch_ms": ms, "achieved": b / (ms * 1e-3) / 1e9, "unit": "GB/s",
                                 "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic.get("score_walk_kernel"),
                                 "traffic_note": "PMC bytes per launch (64-byte sectors, fetch correction 1), profiles/traffic_config2.json"}
        res["score_kernels_ms_per_step"] = {k: round(v[1] / steps, 4) for k, v in prof.items()
                                            if k.startswith(("score_", "kgram_", "query_"))}
    index.close()
    return res


def cpu_baseline(args, synthetic):
    """The oracle (C port of the reference algorithm): one core on the bench document itself, and all host
    cores over the documents of configs[2] (one oracle build per document, a thread each -- the C calls
    release the interpreter lock)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import easa_oracle
    easa_oracle.build()
    n_bytes = int(args.cpu_sample_mib * (1 << 20))
    rng = np.random.default_rng(20240 + 2)                 # (rank 0's seed: the bench document itself at 64 MiB)
    if args.mode == "text":
        _, symbols, m = synthetic.word_stream_document(rng, n_bytes, want_text=False)
    else:
        symbols, m = synthetic.direct_document(rng, n_bytes + 1)
    t0 = time.perf_counter()
    orc = easa_oracle.OracleEASA(symbols=symbols, n_strings=m)
    t_build = time.perf_counter() - t0
    qs, qo = synthetic.keyphrases(rng, symbols, 100)
    t0 = time.perf_counter()
    for k in range(100):
        orc.score_symbols(qs[qo[k]:qo[k + 1]], True, fast=False)     # the reference's own walk (sibling chains)
    t_score = time.perf_counter() - t0
    del orc
    res = {"value": n_bytes / t_build, "unit": "chars/s", "cores": 1, "kind": "port",
           "sample": "oracle/easa_oracle.c full EASA build (DC3+Kasai+childtab+anntab) of the %g MiB bench document "
                     "(%d symbols, %d strings), %.1f s on one core; host has %d cores"
                     % (args.cpu_sample_mib, symbols.size, m, t_build, os.cpu_count()),
           "build_symbols_per_s": symbols.size / t_build, "keyphrase_scores_per_s": 100 / t_score}
    # all cores over documents (configs[2]: 256 x 1 MiB)
    cores = os.cpu_count() or 1
    rng = np.random.default_rng(20240 + 3)
    docs = [synthetic.word_stream_document(rng, 1 << 20, want_text=False)[1:] for _ in range(256)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as pool:
        list(pool.map(lambda d: easa_oracle.OracleEASA(symbols=d[0], n_strings=d[1]).n, docs))
    t_all = time.perf_counter() - t0
    res["over_documents"] = {"value": 256 * (1 << 20) / t_all, "unit": "chars/s", "cores": cores,
                             "sample": "256 documents of 1 MiB (configs[2]), one oracle build per document on a pool of "
                                       "%d threads, %.1f s" % (cores, t_all)}
    return res


if __name__ == "__main__":
    if "--in-process-child" in sys.argv[1:]:
        in_process_child_main(sys.argv[1:])
    else:
        main()
