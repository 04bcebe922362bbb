#!/bin/bash
# On the GPU box: the default bench (with its configs[2] leg) for the in-tree library and every build/variants/lib_*.so
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in ast-text-analysis_amd/east/_lib/libeast_hip.so build/variants/lib_*.so; do
  [ -f "$lib" ] || continue
  EAST_HIP_LIBRARY="$PWD/$lib" timeout 300 python3 bench.py --full-line --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; c=d['config2']; k2=c['kernels_ms_per_step']
print('$(basename $lib .so)', 'build', round(d['build_ms'],3), 'ann', round(k.get('ann_stream_kernel',0),3), round(k.get('ann_wide_kernel',0),3), '| config2 build', round(c['build_ms'],3), 'ann', round(k2.get('ann_stream_kernel',0),3), round(k2.get('ann_wide_kernel',0),3))
"
done
