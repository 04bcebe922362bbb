#!/bin/bash
# rocprofv3 kernel statistics + the two PMC passes of the configs[2] workload (256 x 1 MiB word-stream documents, 10 000 keyphrases)
# -> gpurun_out/config2/; summarise with  tools/summarize_profiles.py r04_config2 config2 traffic_config2.json
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/config2
rm -rf "$OUT"; mkdir -p "$OUT"
echo "${EAST_COMMIT:-unknown}" > $OUT/commit.txt     # (the tree the passes are taken on: gpurun -- "EAST_COMMIT=$(git rev-parse --short HEAD) tools/...")
ARGS="--docs 256 --doc-mib 1 --keyphrases 10000 --no-cpu-baseline --no-config2 --no-extras"
export EAST_BENCH_DETAIL=$OUT/bench_detail.json
timeout 300 python3 bench.py $ARGS 2>/dev/null | tail -1 > $OUT/bench.json
unset EAST_BENCH_DETAIL
export EAST_BENCH_DETAIL=$OUT/bench_profiled_detail.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS --steps 5 --warmup 2 > $OUT/bench_profiled.json 2>/dev/null
unset EAST_BENCH_DETAIL
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS --steps 2 --warmup 1 > $OUT/fetch.json 2> $OUT/fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py $ARGS --steps 2 --warmup 1 > $OUT/write.json 2> $OUT/write.err
python3 - <<'PY'
import csv, glob, json
b = json.load(open("gpurun_out/config2/bench.json"))
print("step %.2f ms  build %.2f  score %.2f  %.3e chars/s" % (b["ms_per_step"], b["build_ms"], b["score_ms"], b["value"]))
f = glob.glob("gpurun_out/config2/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-90s %5s calls %9.1f us avg %6.2f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
