#!/bin/bash
# On the GPU box: the Zipf stand-in (100 x 1 MiB, bench.py --corpus zipf) under a list of environment settings
# ("name:VAR=1 VAR2=x" ...; "base:" = none).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for spec in "$@"; do
  name=${spec%%:*}; vars=${spec#*:}
  env $vars timeout 300 python3 bench.py --full-line --corpus zipf --docs 100 --doc-mib 1 --keyphrases 1000 --no-cpu-baseline --no-config2 --no-extras --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', 'ZIPF build', round(d['build_ms'],3), 'step', round(d['ms_per_step'],3), 'rounds', d['dc3_refine_rounds'], 'lds_sorted', d['lds_sorted'], 'passes', d['radix_passes'], 'fused', d.get('fused_finish'))
print('   ', [(k.replace('_kernel',''),round(v,3)) for k,v in list(d['kernels_ms_per_step'].items())[:14]])
"
done
