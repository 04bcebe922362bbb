#!/usr/bin/env python3
"""Turn gpurun_out/final/ (tools/profile_round.sh) into the tracked files under profiles/:

  profiles/<tag>_kernel_stats.csv       rocprofv3 --kernel-trace --stats summary (as emitted)
  profiles/<tag>_bench.json             the bench line of the un-profiled run
  profiles/<tag>_bench_profiled.json    the bench line measured under rocprofv3
  profiles/<tag>_pmc_per_kernel.csv     FETCH_SIZE / WRITE_SIZE per kernel and per launch
  profiles/traffic.json                 HBM bytes per launch for bench.py's roofline.traffic

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts a wide coalesced streaming
read at exactly half its bytes (MI355X_MICROARCH.md, HBM section); checked here on kernels
with a known byte count (remap_kernel reads 8 B/symbol, radix_hist_kernel<u64> 8 B/element):
the ratio printed below is ~0.49.  traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 for the
streaming kernels; gather kernels are reported raw as well (their correction lies between 1x and 2x).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"


def short(name):
    name = name.split("(")[0].replace("void ", "")
    return name.replace("unsigned long", "u64").replace("unsigned int", "u32").replace(", 1024", "")


def agg(pattern, counter):
    out = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(pattern):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                k = short(r["Kernel_Name"])
                out[k][0] += 1
                out[k][1] += float(r["Counter_Value"])
    return out


stats = glob.glob(os.path.join(SRC, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(DST, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(SRC, "bench.json"), os.path.join(DST, tag + "_bench.json"))
shutil.copy(os.path.join(SRC, "bench_profiled.json"), os.path.join(DST, tag + "_bench_profiled.json"))
fetch = agg(os.path.join(SRC, "fetch", "*", "*_counter_collection.csv"), "FETCH_SIZE")
write = agg(os.path.join(SRC, "write", "*", "*_counter_collection.csv"), "WRITE_SIZE")
bench = json.load(open(os.path.join(SRC, "bench.json")))
n = bench["config"]["symbols_per_gpu"]
traffic = {}
with open(os.path.join(DST, tag + "_pmc_per_kernel.csv"), "w") as f:
    f.write("kernel,launches,FETCH_SIZE_KiB_per_launch,WRITE_SIZE_KiB_per_launch,"
            "hbm_bytes_per_launch_fetch_x2_plus_write\n")
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0])[1] + write.get(k, [0, 0])[1])):
        fc, fv = fetch.get(k, [0, 0.0])
        wc, wv = write.get(k, [0, 0.0])
        c = max(fc, wc, 1)
        per = (2.0 * fv / max(fc, 1) + wv / max(wc, 1)) * 1024.0
        f.write("%s,%d,%.1f,%.1f,%.0f\n" % (k, c, fv / max(fc, 1), wv / max(wc, 1), per))
        traffic[k] = per
# bench.py's kernel names
names = {"radix_scatter_kernel<u64>": "radix_scatter_kernel<u64>", "radix_scatter_kernel<u32>": "radix_scatter_kernel<u32>",
         "radix_hist_kernel<u64>": "radix_hist_kernel<u64>", "radix_hist_kernel<u32>": "radix_hist_kernel<u32>"}
out = {"_note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes of "
                "`python3 bench.py --steps 2 --warmup 1` (gfx950: FETCH_SIZE counts streaming reads at 1/2)",
       "_workload": bench["config"]["workload"]}
for k, v in traffic.items():
    out[names.get(k, k)] = v
json.dump(out, open(os.path.join(DST, "traffic.json"), "w"), indent=1, sort_keys=True)
for k, expect in (("remap_kernel", 8.0 * n), ("radix_hist_kernel<u64>", None)):
    if k in fetch and expect:
        print("calibration %s: FETCH_SIZE*1024 / known read bytes = %.3f"
              % (k, fetch[k][1] / fetch[k][0] * 1024.0 / expect))
print("wrote", tag, "->", DST)
