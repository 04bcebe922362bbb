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
    // (unconditional reads -- an index behind the end reads the last element and counts as 0 --: all sixteen are requested
    // before the first is added; `if (i < n) sum += in(i)` was a round trip per group of four)
    u32 sum = 0, x[SCAN_IPT];
#pragma unroll
    for (int j = 0; j < SCAN_IPT; j++) {
        const u32 i = base + j * BLOCK + threadIdx.x;     // striped: coalesced
        x[j] = in(i < n ? i : n - 1u);
    }
#pragma unroll
    for (int j = 0; j < SCAN_IPT; j++) sum += base + j * BLOCK + threadIdx.x < n ? x[j] : 0u;
    sum = wave_sum(sum);
    if (lane_id() == 0) lds[wave_id()] = sum;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// four consecutive inputs starting at i (i is a multiple of 4); zeros past n (n > 0).  The reads are unconditional -- an
// index behind the end reads the last element --, so that a kernel can request all its rows before it uses the first:
// behind `i < n ?` every row waited for its own round trip.
template <class In>
__device__ __forceinline__ void scan_load4(const In &in, u32 i, u32 n, u32 x[4])
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32 v = in(i + k < n ? i + k : n - 1u);
        x[k] = i + k < n ? v : 0u;
    }
}

// (arrays come from the 256-byte aligned arena, whose blocks are whole multiples of 256 bytes: the 16-byte group of any
// valid element lies inside its block, also where it reaches behind n)
__device__ __forceinline__ void scan_load4(const ArrIn &in, u32 i, u32 n, u32 x[4])
{
    const uint4 v = *reinterpret_cast<const uint4 *>(in.p + (i < n ? i : 0u));
    x[0] = i < n ? v.x : 0u;
    x[1] = i + 1 < n ? v.y : 0u;
    x[2] = i + 2 < n ? v.z : 0u;
    x[3] = i + 3 < n ? v.w : 0u;
}

// out[i] = (exclusive or inclusive) prefix of in over [0, i], plus the scanned
// sum of all earlier tiles.  block_offsets == nullptr means a single tile.
// Wave w owns 1024 contiguous inputs as 4 rows of 64 lanes x 4 inputs: every
// load and store is a coalesced 1 KiB (16 B per lane) wavefront access.
// raw != 0: block_offsets holds the tiles' SUMS as scan_reduce_kernel wrote them, not their prefix sums -- the workgroup
// adds up the sums of the tiles in front of its own (at most SCAN_RAW_TILES of them: a few coalesced loads per thread),
// which saves the launch that scans the sums: most scans of a build run over a few hundred tiles, where a launch costs
// more than the kernel in it.
#define SCAN_RAW_TILES 2048u
template <class In, bool INCLUSIVE>
__global__ __launch_bounds__(BLOCK) void scan_apply_kernel(In in, u32 n, const u32 *block_offsets,
                                                           u32 *out, int raw = 0)
{
    __shared__ u32 lds[WAVES_PER_BLOCK];
    __shared__ u32 lds_raw[WAVES_PER_BLOCK];
    u32 raw_base = 0;
    if (raw) {
        u32 part = 0, bs[SCAN_RAW_TILES / BLOCK];
#pragma unroll
        for (u32 j = 0; j < SCAN_RAW_TILES / BLOCK; j++) {
            const u32 t = threadIdx.x + j * BLOCK;
            bs[j] = block_offsets[t < blockIdx.x ? t : 0u];
        }
#pragma unroll
        for (u32 j = 0; j < SCAN_RAW_TILES / BLOCK; j++) part += threadIdx.x + j * BLOCK < blockIdx.x ? bs[j] : 0u;
        part = wave_sum(part);
        if (lane_id() == 0) lds_raw[wave_id()] = part;
        __syncthreads();
        raw_base = lds_raw[0] + lds_raw[1] + lds_raw[2] + lds_raw[3];
    }
    const u32 lane = lane_id(), w = wave_id();
    const u32 wave_base = blockIdx.x * SCAN_TILE + w * (SCAN_TILE / WAVES_PER_BLOCK);
    u32 v[4][4];
    u32 carry = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) scan_load4(in, wave_base + r * 256 + lane * 4, n, v[r]);     // (all four rows requested first)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        u32 x[4] = {v[r][0], v[r][1], v[r][2], v[r][3]};
        const u32 t = x[0] + x[1] + x[2] + x[3];
        const u32 inc = wave_inclusive_sum(t);
        u32 run = carry + inc - t;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (INCLUSIVE) run += x[k];
            v[r][k] = run;
            if (!INCLUSIVE) run += x[k];
        }
        carry += __shfl(inc, 63, WAVE);
    }
    if (lane == 0) lds[w] = carry;
    __syncthreads();
    u32 base = raw ? raw_base : block_offsets ? block_offsets[blockIdx.x] : 0u;
    if (w > 0) base += lds[0];
    if (w > 1) base += lds[1];
    if (w > 2) base += lds[2];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const u32 i = wave_base + r * 256 + lane * 4;
        if (i + 3 < n) {
            *reinterpret_cast<uint4 *>(out + i) =
                make_uint4(v[r][0] + base, v[r][1] + base, v[r][2] + base, v[r][3] + base);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i + k < n) out[i + k] = v[r][k] + base;
        }
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
    const int raw = nb > 1 && nb <= SCAN_RAW_TILES;
    if (nb > 1) {
        sums = ctx.arena->alloc<u32>(nb);
        LAUNCH(ctx, (scan_reduce_kernel<In>), nb, in, n, sums);
        if (!raw) device_scan<ArrIn, false>(ctx, ArrIn{sums}, nb, sums);
    }
    LAUNCH(ctx, (scan_apply_kernel<In, INCLUSIVE>), nb, in, n, (const u32 *)sums, out, raw);
    ctx.arena->release(mark);
}

// ---- two sums at once, over a length the DEVICE knows ------------------------------------------------------------------
// out_a / out_b = exclusive prefix sums of a[] / b[] over the elements [0, min(n, *n_dev + extra)); tiles wholly behind
// that bound return at once and their outputs stay unwritten (nobody reads them).  The streamed text preparation scans
// per-token quantities this way: its kernels run on an upper bound of the token count (half the code points), the real
// count -- a third to a quarter of that -- is on the device, and the two quantities (kept tokens, kept symbols) used to be
// two scans of three launches each over the whole upper bound.
__global__ __launch_bounds__(BLOCK) void scan_reduce2_kernel(const u32 *__restrict__ a, const u32 *__restrict__ b, u32 n,
                                                             const u32 *__restrict__ n_dev, u32 extra, uint2 *__restrict__ block_sums)
{
    __shared__ u32 lds[2 * WAVES_PER_BLOCK];
    const u32 bound = min(n, *n_dev + extra);
    const u32 base = blockIdx.x * SCAN_TILE;
    if (base >= bound) { if (threadIdx.x == 0) block_sums[blockIdx.x] = uint2{0u, 0u}; return; }
    u32 sa = 0, sb = 0;
    u32 x[SCAN_IPT / 4][4], y[SCAN_IPT / 4][4];
#pragma unroll
    for (int j = 0; j < SCAN_IPT / 4; j++) {
        const u32 i = base + (j * BLOCK + threadIdx.x) * 4u;
        scan_load4(ArrIn{a}, i, bound, x[j]);
        scan_load4(ArrIn{b}, i, bound, y[j]);
    }
#pragma unroll
    for (int j = 0; j < SCAN_IPT / 4; j++) {
        sa += x[j][0] + x[j][1] + x[j][2] + x[j][3];
        sb += y[j][0] + y[j][1] + y[j][2] + y[j][3];
    }
    sa = wave_sum(sa);
    sb = wave_sum(sb);
    if (lane_id() == 0) { lds[wave_id()] = sa; lds[WAVES_PER_BLOCK + wave_id()] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = uint2{lds[0] + lds[1] + lds[2] + lds[3], lds[4] + lds[5] + lds[6] + lds[7]};
}

// exclusive scan of the tile sums, in place; one workgroup
__global__ __launch_bounds__(BLOCK) void scan_sums2_kernel(uint2 *__restrict__ sums, u32 nb)
{
    __shared__ u32 lds4[WAVES_PER_BLOCK];
    u32 run_a = 0, run_b = 0;
    for (u32 t0 = 0; t0 < nb; t0 += BLOCK) {
        const u32 t = t0 + threadIdx.x;
        const uint2 v = t < nb ? sums[t] : uint2{0u, 0u};
        u32 tot_a, tot_b;
        const u32 ea = block_exclusive_sum(v.x, lds4, tot_a);
        const u32 eb = block_exclusive_sum(v.y, lds4, tot_b);
        if (t < nb) sums[t] = uint2{run_a + ea, run_b + eb};
        run_a += tot_a;
        run_b += tot_b;
    }
}

__global__ __launch_bounds__(BLOCK) void scan_apply2_kernel(const u32 *__restrict__ a, const u32 *__restrict__ b, u32 n,
                                                            const u32 *__restrict__ n_dev, u32 extra,
                                                            const uint2 *__restrict__ block_offsets, u32 *__restrict__ out_a,
                                                            u32 *__restrict__ out_b)
{
    __shared__ u32 lds[2 * WAVES_PER_BLOCK];
    const u32 bound = min(n, *n_dev + extra);
    if (blockIdx.x * SCAN_TILE >= bound) return;
    const u32 lane = lane_id(), w = wave_id();
    const u32 wave_base = blockIdx.x * SCAN_TILE + w * (SCAN_TILE / WAVES_PER_BLOCK);
    u32 va[4][4], vb[4][4];
    u32 carry_a = 0, carry_b = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const u32 i = wave_base + r * 256 + lane * 4;
        scan_load4(ArrIn{a}, i, bound, va[r]);
        scan_load4(ArrIn{b}, i, bound, vb[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        u32 x[4] = {va[r][0], va[r][1], va[r][2], va[r][3]}, y[4] = {vb[r][0], vb[r][1], vb[r][2], vb[r][3]};
        const u32 ta = x[0] + x[1] + x[2] + x[3], tb = y[0] + y[1] + y[2] + y[3];
        const u32 ia = wave_inclusive_sum(ta), ib = wave_inclusive_sum(tb);
        u32 ra = carry_a + ia - ta, rb = carry_b + ib - tb;
#pragma unroll
        for (int k = 0; k < 4; k++) { va[r][k] = ra; ra += x[k]; vb[r][k] = rb; rb += y[k]; }
        carry_a += __shfl(ia, 63, WAVE);
        carry_b += __shfl(ib, 63, WAVE);
    }
    if (lane == 0) { lds[w] = carry_a; lds[WAVES_PER_BLOCK + w] = carry_b; }
    __syncthreads();
    const uint2 off = block_offsets ? block_offsets[blockIdx.x] : uint2{0u, 0u};
    u32 base_a = off.x, base_b = off.y;
    for (u32 q = 0; q < w; q++) { base_a += lds[q]; base_b += lds[WAVES_PER_BLOCK + q]; }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const u32 i = wave_base + r * 256 + lane * 4;
        if (i + 3 < bound) {
            *reinterpret_cast<uint4 *>(out_a + i) = make_uint4(va[r][0] + base_a, va[r][1] + base_a, va[r][2] + base_a, va[r][3] + base_a);
            *reinterpret_cast<uint4 *>(out_b + i) = make_uint4(vb[r][0] + base_b, vb[r][1] + base_b, vb[r][2] + base_b, vb[r][3] + base_b);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i + k < bound) { out_a[i + k] = va[r][k] + base_a; out_b[i + k] = vb[r][k] + base_b; }
        }
    }
}

// a, b, out_a, out_b: arrays of n words from the arena (16-byte aligned); n_dev: one word on the device
static void device_scan_pair_bounded(Ctx &ctx, const u32 *a, const u32 *b, u32 n, const u32 *n_dev, u32 extra, u32 *out_a, u32 *out_b)
{
    if (n == 0) return;
    const u32 nb = ceil_div_u32(n, SCAN_TILE);
    const size_t mark = ctx.arena->mark();
    uint2 *sums = nullptr;
    if (nb > 1) {
        sums = ctx.arena->alloc<uint2>(nb);
        LAUNCH(ctx, scan_reduce2_kernel, nb, a, b, n, n_dev, extra, sums);
        LAUNCH(ctx, scan_sums2_kernel, 1, sums, nb);
    }
    LAUNCH(ctx, scan_apply2_kernel, nb, a, b, n, n_dev, extra, (const uint2 *)sums, out_a, out_b);
    ctx.arena->release(mark);
}
