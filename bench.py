#!/usr/bin/env python3
"""bench.py -- the EAST hot path on N MI355X GPUs (one process per GPU).

Workload (BASELINE.json configs[1], one copy per GPU = weak scaling):
  one synthetic 64 MiB random-ASCII word-stream document per GPU, turned into
  3-word strings exactly as `east keyphrases table` does (text mode), and 1 000
  keyphrases.  The corpus shards at document granularity: rank r owns one AST
  shard; the K x N score table is assembled with one RCCL all-gather.

A "step" is one pass of the whole hot path over that batch, with the symbol
stream already resident in HBM:
    EASA build (dense remap, DC3 suffix array, LCP, annotation + child tables)
    + score table (K keyphrases x local documents) [+ all-gather of the blocks]
value = input document bytes of all ranks / step time.

The JSON line also carries
  roofline      -- the dominant kernel of the timed region (by summed HIP-event
                   time on the library's own stream): algorithmic bytes / time
  cpu_baseline  -- the CPU oracle (C port of the reference's easa.py) timed on
                   this box's host cores on a bounded sample (rank 0, N=1 only)

Launch:  python bench.py [--gpus N --steps K --warmup W]
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--doc-mib", type=float, default=64.0, help="size of one document in MiB")
    ap.add_argument("--docs", type=int, default=1, help="documents per GPU (BASELINE configs[2]: 256 x 1 MiB)")
    ap.add_argument("--keyphrases", type=int, default=1000)
    ap.add_argument("--mode", choices=["text", "direct"], default="text",
                    help="text: 3-word strings as the CLI does; direct: get_ast([one string])")
    ap.add_argument("--corpus", choices=["words", "zipf"], default="words",
                    help="words: uniform A-Z word stream (configs 1-3); zipf: natural-language-like (config 5)")
    ap.add_argument("--duplicate-docs", type=int, default=0,
                    help="the last N documents are copies of the first N (long repeats across documents)")
    ap.add_argument("--denormalized", action="store_true", help="the CLI's -d")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=float, default=32.0,
                    help="size of the CPU-baseline sample document (about 10-30 s of single-core work)")
    return ap.parse_args()


def kernel_bytes(name, info, n, n_q, n_docs):
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
        # 4 B LCP read + 4 B annotation write per rank
        "ann_kernel": n * 8,
        # 4 B sorted sample read + one random 4 B rank store per sample
        "dc3_rank_kernel": info["merge_elements"] * 2 // 3 * 8,
        # 8 symbols + 2 ranks read, one 16 B record written per symbol
        "dc3_records_kernel": n * 26,
    }
    return table.get(name)


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

    # ---- synthetic inputs (seed = 20240 + config# + rank), outside the timed region ----
    doc_bytes = int(args.doc_mib * (1 << 20))
    n_bytes = doc_bytes * args.docs                       # input bytes per GPU
    rng = np.random.default_rng(20240 + 2 + 1000 * rank)
    parts, ms = [], []
    vocab = synthetic.zipf_vocabulary(np.random.default_rng(20245)) if args.corpus == "zipf" else None
    for _ in range(args.docs):
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
    m = int(sum(ms))
    n = int(symbols.size)
    D = args.docs
    K = args.keyphrases
    # keyphrases: every rank contributes K/world sampled from its own document, the set is replicated
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
    doc_offsets = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.int64)
    n_strings = np.array(ms, dtype=np.int32)
    local_block = torch.empty((K, D), dtype=torch.float64, device=dev)    # K x D_local
    full_table = torch.empty((world * K, D), dtype=torch.float64, device=dev) if use_dist else None

    index = hip_backend.HipIndex(local_rank, reserve_symbols=n)
    index.build_device(d_symbols.data_ptr(), n, doc_offsets, n_strings)    # also sizes the score scratch
    index.set_keyphrases(q_symbols, q_offsets)

    def step():
        index.build_device(d_symbols.data_ptr(), n, doc_offsets, n_strings)
        index.score_resident(not args.denormalized, local_block.data_ptr())
        if use_dist:
            dist.all_gather_into_tensor(full_table, local_block)           # RCCL over xGMI

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    index.profile_enable(True)
    build_ms, score_ms = [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        build_ms.append(index.last_build_ms)
        score_ms.append(index.last_score_ms)
    fence()
    elapsed = time.perf_counter() - t0
    prof = index.profile_report()
    index.profile_enable(False)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    info = index.info()

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * n_bytes / (elapsed / args.steps)
        # ---- roofline of the dominant kernel (HIP events on the library's stream) ----
        total_kernel_ms = sum(ms for _, ms in prof.values())
        dom = max(prof.items(), key=lambda kv: kv[1][1])
        dom_name, (dom_launches, dom_ms) = dom
        bytes_per_step = kernel_bytes(dom_name, info, n, int(q_offsets[-1]), D)
        roofline = {"kernel": dom_name, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "launches_per_step": dom_launches // args.steps,
                    "avg_launch_ms": dom_ms / dom_launches,
                    "share_of_kernel_time": dom_ms / total_kernel_ms, "traffic": None}
        if bytes_per_step is not None:
            achieved = bytes_per_step * args.steps / (dom_ms * 1e-3) / 1e9
            roofline.update({"achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": bytes_per_step / (dom_launches // args.steps)})
        else:
            roofline.update({"achieved": None, "frac": None})
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        traffic = {}
        if os.path.exists(traffic_file):
            with open(traffic_file) as f:
                traffic = json.load(f)
        roofline["traffic"] = traffic.get(dom_name)
        # the same accounting for every kernel with a byte model, largest first (the time is spread
        # over several kernels of two kinds: streaming sort passes and random-sector gathers)
        by_kernel = []
        for name, (launches, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]:
            b = kernel_bytes(name, info, n, int(q_offsets[-1]), D)
            entry = {"kernel": name, "ms_per_step": ms / args.steps, "share_of_kernel_time": ms / total_kernel_ms}
            if b is not None:
                gbs = b * args.steps / (ms * 1e-3) / 1e9
                entry.update({"achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "traffic": traffic.get(name)})
            by_kernel.append(entry)
        out = {
            "metric": "corpus chars/sec (SA+annotation build + keyphrase score table)",
            "value": value, "unit": "chars/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (symbols/indices) + f64 (scores)", "data": "synthetic",
            "config": {"workload": "%d synthetic %g MiB %s doc(s) per GPU (%s mode), "
                                   "%d keyphrases, %s scores, easa-HIP"
                                   % (D, args.doc_mib, "random-ASCII word-stream" if vocab is None else
                                      "Zipf natural-language-like", args.mode, K,
                                      "denormalized" if args.denormalized else "normalized"),
                       "symbols_per_gpu": n, "strings_per_gpu": m, "parallelism": "doc-shard x%d" % world},
            "build_ms": float(np.mean(build_ms)), "score_ms": float(np.mean(score_ms)),
            "build_symbols_per_s": n / (float(np.mean(build_ms)) * 1e-3),
            "build_chars_per_s": n_bytes / (float(np.mean(build_ms)) * 1e-3),
            "keyphrase_scores_per_s": K * D / (float(np.mean(score_ms)) * 1e-3),
            "build_algorithmic_GBps": 16.0 * n / (float(np.mean(build_ms)) * 1e-3) / 1e9,
            "dc3_levels": info["dc3_levels"], "dc3_levels_resolved": info["dc3_levels_resolved"], "dc3_refine_rounds": info["refine_rounds"], "window_sorted": info["window_sorted"],
            "radix_passes": info["radix_passes"],
            "roofline": roofline, "roofline_by_kernel": by_kernel,
            "kernels_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in
                                    sorted(prof.items(), key=lambda kv: -kv[1][1])},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, synthetic)
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
        print(json.dumps(out), flush=True)


def cpu_baseline(args, synthetic):
    """The oracle (C port of the reference algorithm, single thread) on a bounded sample."""
    from oracle import easa_oracle
    easa_oracle.build()
    n_bytes = int(args.cpu_sample_mib * (1 << 20))
    rng = np.random.default_rng(20240 + 2)
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
    return {"value": n_bytes / t_build, "unit": "chars/s", "cores": 1, "kind": "port",
            "sample": "oracle/easa_oracle.c full EASA build (DC3+Kasai+childtab+anntab) of one %g MiB "
                      "word-stream doc (%d symbols, %d strings), %.1f s; host has %d cores"
                      % (args.cpu_sample_mib, symbols.size, m, t_build, os.cpu_count()),
            "build_symbols_per_s": symbols.size / t_build,
            "keyphrase_scores_per_s": 100 / t_score}


if __name__ == "__main__":
    main()
