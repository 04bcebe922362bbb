#!/usr/bin/env python3
"""Turn gpurun_out/final/ (tools/profile_round.sh) into the tracked files under profiles/:

  profiles/<tag>_kernel_stats.csv       rocprofv3 --kernel-trace --stats summary (as emitted)
  profiles/<tag>_bench.json             the bench line of the un-profiled run
  profiles/<tag>_bench_profiled.json    the bench line measured under rocprofv3
  profiles/<tag>_pmc_per_kernel.csv     FETCH_SIZE / WRITE_SIZE per kernel and per launch
  profiles/traffic.json                 HBM bytes per launch for bench.py's roofline.traffic

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts a wide coalesced streaming
read at exactly half its bytes (MI355X_MICROARCH.md, HBM section); checked here on kernels
with a known byte count (presence_remap_kernel, remap_bytes_kernel and presence_kernel read 4 B/symbol):
the ratio printed below is ~0.50.  traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 for the
streaming kernels.  Gather kernels (GATHER below) issue 64-byte sector requests, which the
counter tallies at their true size, so their FETCH_SIZE is NOT doubled.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03_final"
# summarize_profiles.py <tag> [<directory under gpurun_out/>] [<traffic file name>]
#   r03_final final traffic.json        (tools/profile_round.sh, the default workload)
#   r03_zipf  zipf  traffic_zipf.json   (tools/profile_zipf.sh, the config 5 stand-in)
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "final")
TRAFFIC = sys.argv[3] if len(sys.argv) > 3 else "traffic.json"


# kernels whose reads are dominated by random gathers / whose writes by random scatters
GATHER = ("dc3_merge_tile_kernel", "dc3_merge_partition_kernel", "dc3_merge_lcp_tile_kernel",
          "dc3_merge_partition_rec_kernel", "dc3_lcp_heads_kernel", "lcp8_kernel", "lcp_kernel",
          "dc3_compact_s0_bytes_kernel", "dc3_compact_s0_kernel", "dc3_rank_kernel", "dc3_scatter_names_kernel",
          "dc3_resolve_ties_text_kernel", "dc3_resolve_ties_kernel", "kgram_mark_kernel", "score_walk_kernel",
          "dc3_pair_keys_kernel", "dc3_gather_third_kernel", "ann_wide_kernel", "child_kernel", "doc_keys_kernel",
          # mixed: a streaming pass (counted at 1/2) plus text gathers for the tied suffixes (counted in full);
          # with c = 1 the figure is a lower bound, short by the streamed half (4 B/suffix)
          "dc3_refine_classify_kernel", "lvl0_place_kernel", "lvl0_lcp_keys_kernel", "dc3_refine_keys_kernel",
          # the fused finish: 8 B per suffix streamed in (counted at 1/2: the figure is short by 4 B per suffix) plus one
          # 4-byte text gather per tied suffix (a 64-byte sector, counted in full)
          "lvl0_finish_kernel", "lvl0_lcp_text_kernel",
          # the in-LDS round: 12 B per element streamed (short by 6 B per element) plus a 16-byte text gather each
          "refine_lds_sort_kernel", "dc3_double_keys_kernel")


def short(name):
    """rocprofv3's demangled name -> the name bench.py / east_hip_profile_report use."""
    name = name.split("(")[0].replace("void ", "")
    name = name.replace("unsigned long", "u64").replace("unsigned int", "u32").replace(", 1024", "")
    name = name.replace("> >", ">>")
    name = name.replace(">, false>", ">>").replace(">, true>", ">>")       # (radix_hist_kernel's third parameter: the one-bin test on / off)
    for k in ("u32", "u64"):
        name = name.replace(", PairSrc<%s>>" % k, ">").replace(", WindowSrc<%s>>" % k, ",gen>")
        name = name.replace(", TextWindowGen<%s>>" % k, ",gen>")
    for plain in ("dc3_refine_classify_kernel", "dc3_refine_compact_kernel", "dc3_refine_restore_kernel", "lvl0_place_kernel",
                  "lvl0_lcp_keys_kernel", "validate_n_strings_kernel", "score_walk_kernel", "lvl0_finish_kernel"):
        if name.startswith(plain + "<"):
            name = plain
    return name


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: of several runs' outputs only the last one counts."""
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]


def agg(pattern, counter):
    out = collections.defaultdict(lambda: [0, 0.0])
    for path in newest(pattern):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                k = short(r["Kernel_Name"])
                out[k][0] += 1
                out[k][1] += float(r["Counter_Value"])
    return out


stats = newest(os.path.join(SRC, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(DST, tag + "_kernel_stats.csv"))
# (round 6: bench.py prints a line of under 6 KB for the driver; everything else -- the per-kernel tables the recomputation below
# works on -- is in bench_detail.json, of which the profiling scripts keep one per run)
for kind in ("bench", "bench_profiled"):
    shutil.copy(os.path.join(SRC, kind + ".json"), os.path.join(DST, tag + "_" + kind + ".json"))
    if os.path.exists(os.path.join(SRC, kind + "_detail.json")):
        shutil.copy(os.path.join(SRC, kind + "_detail.json"), os.path.join(DST, tag + "_" + kind + "_detail.json"))
fetch = agg(os.path.join(SRC, "fetch", "*", "*_counter_collection.csv"), "FETCH_SIZE")
write = agg(os.path.join(SRC, "write", "*", "*_counter_collection.csv"), "WRITE_SIZE")
bench = json.loads(open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1])
n = bench["config"]["symbols_per_gpu"]
traffic = {}
with open(os.path.join(DST, tag + "_pmc_per_kernel.csv"), "w") as f:
    f.write("kernel,launches,FETCH_SIZE_KiB_per_launch,WRITE_SIZE_KiB_per_launch,fetch_correction,"
            "hbm_bytes_per_launch\n")
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0])[1] + write.get(k, [0, 0])[1])):
        fc, fv = fetch.get(k, [0, 0.0])
        wc, wv = write.get(k, [0, 0.0])
        c = max(fc, wc, 1)
        corr = 1.0 if k in GATHER else 2.0
        per = (corr * fv / max(fc, 1) + wv / max(wc, 1)) * 1024.0
        f.write("%s,%d,%.1f,%.1f,%.0f,%.0f\n" % (k, c, fv / max(fc, 1), wv / max(wc, 1), corr, per))
        traffic[k] = per
# bench.py's kernel names
names = {"radix_scatter_kernel<u64>": "radix_scatter_kernel<u64>", "radix_scatter_kernel<u32>": "radix_scatter_kernel<u32>",
         "radix_hist_kernel<u64>": "radix_hist_kernel<u64>", "radix_hist_kernel<u32>": "radix_hist_kernel<u32>"}
out = {"_note": "HBM bytes per launch = (c*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes of "
                "`python3 bench.py --steps 2 --warmup 1`; c = 2 for streaming kernels (gfx950 counts wide streaming "
                "reads at 1/2, calibrated on remap_kernel), c = 1 for gather kernels (64-byte sector requests)",
       "_fetch_correction_1": "kernels priced with c = 1 (a LOWER bound where part of their reads is 16-byte streaming: "
                              "lvl0_finish_kernel / lvl0_place_kernel are short by about 4 B per suffix, the refinement "
                              "rounds' kernels by half of their streamed reads): " + ", ".join(sorted(k for k in traffic if k in GATHER)),
       "_workload": bench["config"]["workload"],
       "_commit": (open(os.path.join(SRC, "commit.txt")).read().strip() if os.path.exists(os.path.join(SRC, "commit.txt")) else None)}
for k, v in traffic.items():
    out[names.get(k, k)] = v
json.dump(out, open(os.path.join(DST, TRAFFIC), "w"), indent=1, sort_keys=True)
# the bench lines of this very run carry the PMC traffic of this run (bench.py itself reads the
# traffic.json that was committed before it started)
for name in (tag + "_bench.json", tag + "_bench_profiled.json"):
    path = os.path.join(DST, name)
    detail_path = path[:-len(".json")] + "_detail.json"
    has_detail = os.path.exists(detail_path)
    line = json.load(open(detail_path)) if has_detail else json.loads(open(path).read().strip().splitlines()[-1])
    line["roofline"]["traffic"] = out.get(line["roofline"]["kernel"])
    for e in line.get("roofline_by_kernel", []):
        if "traffic" in e or e["kernel"] in out:
            e["traffic"] = out.get(e["kernel"])
    # the whole-build figure with THIS run's counters (bench.py computed it with the traffic.json of the run before)
    launches, ms = line.get("kernel_launches_per_step"), line.get("kernels_ms_per_step")
    if launches and ms and isinstance(line["roofline"].get("rocprof_hbm_fraction"), dict):
        sys.path.insert(0, ROOT)
        import bench
        num = den = 0.0
        for k, t in ms.items():
            if k.startswith(bench.BUILD_KERNEL_PREFIXES) and k in out and k in launches:
                num += out[k] * launches[k]
                den += t * 1e-3
        if den:
            build_ms = sum(t for k, t in ms.items() if k.startswith(bench.BUILD_KERNEL_PREFIXES))
            n_cov = sum(1 for k in ms if k.startswith(bench.BUILD_KERNEL_PREFIXES) and k in out and k in launches)
            line["roofline"]["rocprof_hbm_fraction"].update({"GBps": num / den / 1e9, "frac": num / den / 1e9 / bench.HBM_PEAK_GBS,
                                                             "kernels": n_cov, "share_of_build_kernel_time": den * 1e3 / build_ms})
    if has_detail:                                       # the full record, and the driver's line made from it again
        sys.path.insert(0, ROOT)
        import bench as bench_module
        with open(detail_path, "w") as f:
            json.dump(line, f, indent=1)
        with open(path, "w") as f:
            f.write(bench_module.compact_line(line, detail_path) + "\n")
    else:
        with open(path, "w") as f:
            f.write(json.dumps(line) + "\n")
for k, expect in (("remap_kernel", 8.0 * n), ("remap_bytes_kernel", 4.0 * n), ("presence_kernel", 4.0 * n),
                  ("presence_remap_kernel", 4.0 * n)):
    if k in fetch and expect:
        print("calibration %s: FETCH_SIZE*1024 / known read bytes = %.3f"
              % (k, fetch[k][1] / fetch[k][0] * 1024.0 / expect))
print("wrote", tag, "->", DST)
