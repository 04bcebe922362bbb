#!/bin/bash
# On the GPU box: the score walk per document for collections of 8 .. 256 documents of 1 MiB (10 000 keyphrases) -- does a
# collection whose index fits the 256 MB Infinity Cache walk faster per document than one that does not?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for docs in ${DOCS:-8 16 24 32 64 128 256}; do
  env "$@" timeout 300 python3 bench.py --full-line --docs $docs --doc-mib 1 --keyphrases 10000 --no-cpu-baseline --no-config2 --no-extras --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('docs', $docs, 'build', round(d['build_ms'],3), 'score', round(d['score_ms'],3), 'walk', round(k.get('score_walk_kernel',0),4), 'walk us/doc', round(1e3*k.get('score_walk_kernel',0)/$docs,3))
"
done
