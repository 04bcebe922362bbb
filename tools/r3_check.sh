#!/bin/bash
# On the GPU box: gpu parity tests, then the default bench with and without the fused finish.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r3}
mkdir -p $OUT
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for mode in fused unfused; do
  if [ $mode = unfused ]; then export EAST_HIP_NO_FUSED_FINISH=1; else unset EAST_HIP_NO_FUSED_FINISH; fi
  timeout 300 python3 bench.py --full-line --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:---no-config2 --no-extras} > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  python3 - $OUT/bench_$mode.json $mode <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print(sys.argv[2], "FAILED", e); sys.exit(0)
k = d.get("kernels_ms_per_step", {})
top = sorted(k.items(), key=lambda kv: -kv[1])[:12]
print("%-8s build %.3f step %.3f | %s" % (sys.argv[2], d["build_ms"], d["ms_per_step"], "  ".join("%s=%.3f" % (n.replace("radix_", "").replace("_kernel", ""), v) for n, v in top)))
for key in ("config2", "config5"):
    if key in d: print("   ", key, {kk: vv for kk, vv in d[key].items() if isinstance(vv, (int, float))})
PY
done
