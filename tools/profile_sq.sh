#!/bin/bash
# SQ / LDS / TA / L2 counter passes of one bench.py workload (run on the GPU box through gpurun):
#   tools/profile_sq.sh <name> <bench.py arguments ...>   ->  gpurun_out/<name>_sq/<pass>/...
# e.g.  tools/profile_sq.sh zipf --corpus zipf --docs 100 --doc-mib 1 --keyphrases 1000
#       tools/profile_sq.sh config2 --docs 256 --doc-mib 1 --keyphrases 10000
# Every pass is its own run with --kernel-trace only (counters are never combined with the runtime trace domains),
# rocprofv3 directly in front of `python3 bench.py`.  A pass whose counters do not fit the block's slots fails by itself
# and is simply missing from the summary (tools/summarize_sq.py <name> <tag>).
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
NAME=$1; shift
OUT=gpurun_out/${NAME}_sq
rm -rf "$OUT"; mkdir -p "$OUT"
echo "${EAST_COMMIT:-unknown}" > $OUT/commit.txt
ARGS="$* --no-cpu-baseline --no-config2 --no-extras --steps 2 --warmup 1"
pass() {
  local tag=$1; shift
  timeout 420 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$tag -- python3 bench.py $ARGS > $OUT/$tag.json 2> $OUT/$tag.err \
    || echo "pass $tag failed (rc $?)"
}
# where the waves' cycles go: parked at a wait / barrier, stalled at issue, issuing; resident waves
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LEVEL_WAVES
# LDS: instructions, array cycles, bank / address conflicts; vector and scalar ALU instructions
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
# vector memory: instructions issued (address-divergent loads show as many instructions per byte), their issue cycles
pass sq3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_ATOMIC
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum
# L2: hits / misses (all XCDs summed)
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
ls $OUT; for e in $OUT/*.err; do tail -n 2 "$e" | cut -c1-300; done
