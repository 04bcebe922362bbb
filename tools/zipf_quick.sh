#!/bin/bash
# On the GPU box: the Zipf stand-in (100 x 1 MiB) through bench.py for the in-tree library and every build/variants/lib_*.so
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in ast-text-analysis_amd/east/_lib/libeast_hip.so build/variants/lib_*.so; do
  [ -f "$lib" ] || continue
  EAST_HIP_LIBRARY="$PWD/$lib" timeout 300 python3 bench.py --corpus zipf --docs 100 --doc-mib 1 --keyphrases 1000 --no-cpu-baseline --no-config2 --no-extras --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$(basename $lib .so)', 'build', round(d['build_ms'],3), 'step', round(d['ms_per_step'],3), 'first', round(d.get('first_build_ms',0),3), 'rounds', d['dc3_refine_rounds'], 'lds_sorted', d['lds_sorted'])
print('   ', [(k.replace('_kernel',''),round(v,3)) for k,v in list(d['kernels_ms_per_step'].items())[:12]])
"
done
