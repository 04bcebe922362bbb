#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + the two PMC passes of the
# default bench.py workload, into gpurun_out/final/.  Summarise with tools/summarize_profiles.py.
# PMC passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined
# with --sys-trace etc.), each with the program itself after `--`.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/final
rm -rf "$OUT"; mkdir -p "$OUT"
echo "${EAST_COMMIT:-unknown}" > $OUT/commit.txt     # (the tree the passes are taken on: gpurun -- "EAST_COMMIT=$(git rev-parse --short HEAD) tools/...")
export EAST_BENCH_DETAIL=$OUT/bench_profiled_detail.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 1 --no-config2 --no-extras --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/trace.err
unset EAST_BENCH_DETAIL
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config2 --no-extras > $OUT/fetch.json 2> $OUT/fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config2 --no-extras > $OUT/write.json 2> $OUT/write.err
export EAST_BENCH_DETAIL=$OUT/bench_detail.json
timeout 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
unset EAST_BENCH_DETAIL
find $OUT -name "*.csv" | head -20
tail -c 400 $OUT/bench.json
