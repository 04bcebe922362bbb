// scan.h -- device-wide prefix sums of uint32 (reduce / scan-of-sums / apply).
//
// Three streaming kernels instead of a decoupled look-back single pass: the
// look-back protocol needs agent-scope release/acquire between workgroups on
// different XCDs (per-XCD L2s are not coherent), which costs more than the
// extra 4 B/element read at the sizes used here.  The input is a functor so
// "flags" never have to be materialised in HBM (naming in DC3, terminator
// numbering, S0 compaction all scan a predicate of another array).
#pragma once
#include "common.h"

#define SCAN_IPT 16
#define SCAN_TILE (BLOCK * SCAN_IPT)   // 4096 elements per workgroup

struct ArrIn {             // plain array
    const u32 *p;
    __device__ __forceinline__ u32 operator()(u32 i) const { return p[i]; }
};

template <class In>
__global__ __launch_bounds__(BLOCK) void scan_reduce_kernel(In in, u32 n, u32 *block_sums)
{
    __shared__ u32 lds[WAVES_PER_BLOCK];
    const u32 base = blockIdx.x * SCAN_TILE;
    u32 sum = 0;
#pragma unroll 4
    for (int j = 0; j < SCAN_IPT; j++) {
        const u32 i = base + j * BLOCK + threadIdx.x;     // striped: coalesced
        if (i < n) sum += in(i);
    }
    sum = wave_sum(sum);
    if (lane_id() == 0) lds[wave_id()] = sum;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// out[i] = (exclusive or inclusive) prefix of in over [0, i], plus the scanned
// sum of all earlier tiles.  block_offsets == nullptr means a single tile.
template <class In, bool INCLUSIVE>
__global__ __launch_bounds__(BLOCK) void scan_apply_kernel(In in, u32 n, const u32 *block_offsets,
                                                           u32 *out)
{
    __shared__ u32 lds[WAVES_PER_BLOCK];
    const u32 base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_IPT;   // blocked
    u32 v[SCAN_IPT];
    u32 sum = 0;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; j++) {
        const u32 i = base + j;
        v[j] = i < n ? in(i) : 0u;
        sum += v[j];
    }
    u32 total;
    u32 run = block_exclusive_sum(sum, lds, total);
    if (block_offsets) run += block_offsets[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SCAN_IPT; j++) {
        const u32 i = base + j;
        if (INCLUSIVE) run += v[j];
        if (i < n) out[i] = run;
        if (!INCLUSIVE) run += v[j];
    }
}

// Host driver.  `out` may alias the input array of an ArrIn (every thread
// reads its items before any thread of the same tile writes, tiles are disjoint).
template <class In, bool INCLUSIVE>
static void device_scan(Ctx &ctx, In in, u32 n, u32 *out)
{
    if (n == 0) return;
    const u32 nb = ceil_div_u32(n, SCAN_TILE);
    const size_t mark = ctx.arena->mark();
    u32 *sums = nullptr;
    if (nb > 1) {
        sums = ctx.arena->alloc<u32>(nb);
        LAUNCH(ctx, (scan_reduce_kernel<In>), nb, in, n, sums);
        device_scan<ArrIn, false>(ctx, ArrIn{sums}, nb, sums);
    }
    LAUNCH(ctx, (scan_apply_kernel<In, INCLUSIVE>), nb, in, n, (const u32 *)sums, out);
    ctx.arena->release(mark);
}
