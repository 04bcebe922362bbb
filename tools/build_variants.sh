#!/bin/bash
# Build several copies of libeast_hip.so with different -D flags into build/variants/ (git-ignored, but
# they travel to the GPU box) for A/B timing:  tools/build_variants.sh "a:-DFOO=1" "b:-DFOO=2"
# then on the box:  EAST_HIP_LIBRARY=build/variants/lib_a.so python bench.py --full-line ...
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/build/variants"
BASE="-O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function"
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  ( /opt/rocm/bin/hipcc $BASE $f -o "$ROOT/build/variants/lib_$n.so" "$ROOT/ast-text-analysis_amd/csrc/east_hip.hip" 2>&1 | grep -E "error|warning" ) &
done
wait
ls -la "$ROOT/build/variants/"
