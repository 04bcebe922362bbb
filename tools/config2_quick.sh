#!/bin/bash
# On the GPU box: BASELINE configs[2] (256 x 1 MiB word-stream documents, 10 000 keyphrases) through bench.py for the in-tree
# library under a list of environment settings ("name:VAR=1 VAR2=x" ...; "base:" = none) and for every build/variants/lib_*.so.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {
  local name=$1 lib=$2; shift 2
  env "$@" EAST_HIP_LIBRARY="$PWD/$lib" timeout 300 python3 bench.py --full-line --docs 256 --doc-mib 1 --keyphrases 10000 --no-cpu-baseline --no-config2 --no-extras --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', 'CONFIG2 build', round(d['build_ms'],3), 'score', round(d['score_ms'],3), 'step', round(d['ms_per_step'],3))
print('   ', [(k.replace('_kernel',''),round(v,3)) for k,v in list(d['kernels_ms_per_step'].items())[:16]])
"
}
for spec in "$@"; do
  name=${spec%%:*}; vars=${spec#*:}
  run "$name" ast-text-analysis_amd/east/_lib/libeast_hip.so $vars
done
for lib in build/variants/lib_*.so; do
  [ -f "$lib" ] || continue
  run "$(basename $lib .so)" "$lib"
done
