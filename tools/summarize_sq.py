#!/usr/bin/env python3
"""gpurun_out/<name>_sq/ (tools/profile_sq.sh) -> profiles/<tag>_sq_per_kernel.csv: per kernel, every collected counter
summed over its launches and dimensions, plus the ratios one reads them for:

  wait_any_frac      SQ_WAIT_ANY / SQ_WAVE_CYCLES        waves parked at s_waitcnt or a barrier
  wait_inst_frac     SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   waves stalled at issue (of it wait_inst_lds_frac: the LDS queue)
  active_frac        SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES waves issuing
  waves_per_simd     SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / 7.5   waves resident per SIMD while the kernel runs (SQ_LEVEL_WAVES reads 0 on this
                     pool; the ratio of the two cycle counters is calibrated on kernels whose residency the launch fixes:
                     score_walk_kernel and radix_scatter_kernel, 8 per SIMD by their registers / LDS, give 59.6-60.9 -> the
                     factor 7.5; lvl0_finish_kernel, 6 per SIMD at 78 VGPRs, gives 45.1 -> 6.0)
  lds_conflict_frac  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE   extra LDS-array cycles through bank conflicts
  vmem_rd_per_wave   SQ_INSTS_VMEM_RD / SQ_WAVES         vector loads a wave issues
  l2_hit_frac        TCC_HIT / (TCC_HIT + TCC_MISS)

usage: summarize_sq.py <name under gpurun_out, without _sq> <tag, e.g. r05_zipf>"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
name, tag = sys.argv[1], sys.argv[2]
SRC = os.path.join(ROOT, "gpurun_out", name + "_sq")


def short(k):
    k = k.split("(")[0].replace("void ", "")
    k = k.replace("unsigned long", "u64").replace("unsigned int", "u32").replace("unsigned char", "u8")
    for plain in ("dc3_refine_classify_kernel", "dc3_refine_compact_kernel", "lvl0_place_kernel", "lvl0_finish_kernel",
                  "score_walk_kernel", "refine_lds_sort_kernel", "validate_n_strings_kernel", "lvl0_lcp_keys_kernel"):
        if k.startswith(plain + "<"):
            return plain
    for key in ("u32", "u64"):
        k = k.replace(", PairSrc<%s> >" % key, ">").replace(", TextWindowGen<%s> >" % key, ",gen>").replace(", HtWindowGen<%s> >" % key, ",htgen>")
    return k[:80]


sums = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(int)
counters = []
for pass_dir in sorted(glob.glob(os.path.join(SRC, "*/"))):
    files = sorted(glob.glob(os.path.join(pass_dir, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
    for path in files:
        seen = set()
        for r in csv.DictReader(open(path)):
            k, c = short(r["Kernel_Name"]), r["Counter_Name"]
            sums[k][c] += float(r["Counter_Value"])
            if c not in counters:
                counters.append(c)
            if pass_dir.rstrip("/").endswith("sq1") and c == "SQ_WAVES":
                launches[k] += 1


def ratio(row, a, b):
    r = row[a] / row[b] if a in row and b in row and row[b] else None
    return r / 7.5 if r is not None and (a, b) == ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES") else r


derived = [("wait_any_frac", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), ("wait_inst_frac", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
           ("wait_inst_lds_frac", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"), ("active_frac", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"),
           ("waves_per_simd", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"), ("lds_conflict_frac", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
           ("vmem_rd_per_wave", "SQ_INSTS_VMEM_RD", "SQ_WAVES"), ("vmem_wr_per_wave", "SQ_INSTS_VMEM_WR", "SQ_WAVES"),
           ("lds_inst_per_wave", "SQ_INSTS_LDS", "SQ_WAVES"), ("valu_inst_per_wave", "SQ_INSTS_VALU", "SQ_WAVES"),
           ("ta_busy_per_wave_cycle", "TA_TA_BUSY_sum", "SQ_WAVE_CYCLES")]
dst = os.path.join(ROOT, "profiles", tag + "_sq_per_kernel.csv")
order = sorted(sums, key=lambda k: -sums[k].get("SQ_WAVE_CYCLES", 0.0))
with open(dst, "w") as f:
    f.write("# commit %s; rocprofv3 --pmc passes of `python3 bench.py <%s workload> --steps 2 --warmup 1` (tools/profile_sq.sh); counters "
            "summed over the run's launches of a kernel; SQ cycle counters in quad-cycles\n"
            % (open(os.path.join(SRC, "commit.txt")).read().strip() if os.path.exists(os.path.join(SRC, "commit.txt")) else "unknown", name))
    f.write(",".join(["kernel", "launches"] + [d[0] for d in derived] + ["l2_hit_frac"] + counters) + "\n")
    for k in order:
        row = sums[k]
        vals = [ratio(row, a, b) for _, a, b in derived]
        hit = row.get("TCC_HIT_sum"), row.get("TCC_MISS_sum")
        vals.append(hit[0] / (hit[0] + hit[1]) if hit[0] is not None and hit[1] is not None and hit[0] + hit[1] else None)
        f.write(",".join(['"%s"' % k, str(launches.get(k, 0))] + ["" if v is None else "%.4g" % v for v in vals] +
                         ["%.6g" % row[c] if c in row else "" for c in counters]) + "\n")
print("wrote", dst)
for k in order[:14]:
    row = sums[k]
    print("%-44s" % k[:44], " ".join("%s=%s" % (d[0][:14], "-" if ratio(row, d[1], d[2]) is None else "%.3f" % ratio(row, d[1], d[2])) for d in derived[:7]))
