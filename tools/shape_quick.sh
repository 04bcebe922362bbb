#!/bin/bash
# On the GPU box: bench.py on a word-stream shard of D documents of M MiB under a list of environment settings
# ("name:VAR=1 ..."; "base:" = none).   usage: shape_quick.sh D M spec...
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=$1; M=$2; shift 2
for spec in "$@"; do
  name=${spec%%:*}; vars=${spec#*:}
  env $vars timeout 300 python3 bench.py --full-line --docs $D --doc-mib $M --keyphrases 1000 --no-cpu-baseline --no-config2 --no-extras --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', '$D x $M MiB build', round(d['build_ms'],3), 'score', round(d['score_ms'],3), 'step', round(d['ms_per_step'],3), 'seg', d.get('segmented_sort'), 'passes', d['radix_passes'], 'rounds', d['dc3_refine_rounds'])
print('   ', [(k.replace('_kernel',''),round(v,3)) for k,v in list(d['kernels_ms_per_step'].items())[:12]])
"
done
