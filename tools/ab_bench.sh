#!/bin/bash
# On the GPU box: time bench.py (per-kernel HIP-event profile) for every library in build/variants/
# (and the in-tree one), print the build time and the largest kernels.  Extra args go to bench.py.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/ab
for lib in ast-text-analysis_amd/east/_lib/libeast_hip.so build/variants/lib_*.so; do
  [ -f "$lib" ] || continue
  tag=$(basename "$lib" .so)
  EAST_HIP_LIBRARY="$PWD/$lib" timeout 300 python3 bench.py --full-line --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err
  python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/ab/%s.json" % tag))
except Exception as e:
    print(tag, "FAILED", e); sys.exit(0)
k = d["kernels_ms_per_step"]
top = sorted(k.items(), key=lambda kv: -kv[1])[:9]
print("%-14s build %.3f step %.3f | %s" % (tag, d["build_ms"], d["ms_per_step"], "  ".join("%s=%.3f" % (n.replace("radix_", "").replace("_kernel", ""), v) for n, v in top)))
PY
done
