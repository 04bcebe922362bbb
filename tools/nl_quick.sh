#!/bin/bash
# On the GPU box: the two natural-language legs -- the Zipf stand-in (100 x 1 MiB, bench.py --corpus zipf) and the real
# prose of the image resampled to 64 x 1 MiB (tools/natural_text_bench.py) -- for the in-tree library under a list of
# environment settings ("name:VAR=1 VAR2=x" ...; "base:" = none) and for every build/variants/lib_*.so.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {   # name, lib, env assignments
  local name=$1 lib=$2; shift 2
  env "$@" EAST_HIP_LIBRARY="$PWD/$lib" timeout 300 python3 bench.py --full-line --corpus zipf --docs 100 --doc-mib 1 --keyphrases 1000 --no-cpu-baseline --no-config2 --no-extras --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', 'ZIPF build', round(d['build_ms'],3), 'step', round(d['ms_per_step'],3), 'rounds', d['dc3_refine_rounds'], 'lds_sorted', d['lds_sorted'])
print('   ', [(k.replace('_kernel',''),round(v,3)) for k,v in list(d['kernels_ms_per_step'].items())[:14]])
"
  env "$@" EAST_HIP_LIBRARY="$PWD/$lib" EAST_PROFILE=1 timeout 300 python3 tools/natural_text_bench.py --resample-mib 64 2>/dev/null | grep -E "^build|launches" | head -${PROSE_LINES:-12} | sed "s/^/$name PROSE /"
}
for spec in "$@"; do
  name=${spec%%:*}; vars=${spec#*:}
  run "$name" ast-text-analysis_amd/east/_lib/libeast_hip.so $vars
done
for lib in build/variants/lib_*.so; do
  [ -f "$lib" ] || continue
  run "$(basename $lib .so)" "$lib"
done
