// radix_sort.h -- stable LSD radix sort of (key, value) pairs, 8-bit digits.
//
// One pass = three streaming kernels:
//   radix_hist_kernel     per-tile 256-bin histogram (LDS atomics)      -> hist[digit][tile]
//   device_scan           exclusive scan of the digit-major histogram   -> global bases
//   radix_scatter_kernel  wave64 ballot multisplit for the stable rank inside
//                         the tile, tile re-ordered by digit in LDS, then
//                         written out as contiguous per-digit runs.
//
// Tile = 4096 pairs per 256-thread workgroup: wave w owns the contiguous 1024
// pairs [w*1024, (w+1)*1024) and walks them in 16 rows of 64 (coalesced 64-lane
// loads), so the order (wave, row, lane) is the input order and the sort is
// stable.  LDS: 4096 * (sizeof(K)+4) bytes of staged pairs + 5 KiB of counters
// (u64 keys: 53 KiB -> 2-3 workgroups per CU out of the 160 KiB).
#pragma once
#include "common.h"
#include "scan.h"

#define RS_IPT 16
#define RS_TILE (BLOCK * RS_IPT)       // 4096
#define RS_WAVE_ITEMS (RS_TILE / WAVES_PER_BLOCK)   // 1024
#define RS_BINS 256

template <class K>
__global__ __launch_bounds__(BLOCK) void radix_hist_kernel(const K *__restrict__ keys, u32 n,
                                                           int shift, u32 *__restrict__ hist,
                                                           u32 n_tiles)
{
    __shared__ u32 bins[RS_BINS];
    bins[threadIdx.x] = 0;
    __syncthreads();
    const u32 base = blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int j = 0; j < RS_IPT; j++) {
        const u32 i = base + j * BLOCK + threadIdx.x;
        if (i < n) atomicAdd(&bins[(u32)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * n_tiles + blockIdx.x] = bins[threadIdx.x];
}

template <class K>
__global__ __launch_bounds__(BLOCK) void radix_scatter_kernel(
    const K *__restrict__ keys_in, const u32 *__restrict__ vals_in, K *__restrict__ keys_out,
    u32 *__restrict__ vals_out, u32 n, int shift, const u32 *__restrict__ scanned_hist, u32 n_tiles)
{
    __shared__ K s_keys[RS_TILE];
    __shared__ u32 s_vals[RS_TILE];
    __shared__ u32 wave_cnt[WAVES_PER_BLOCK][RS_BINS];  // per-wave digit counts, then wave bases
    __shared__ u32 digit_start[RS_BINS];                // first slot of the digit inside the tile
    __shared__ u32 global_base[RS_BINS];                // output index of slot 0 of the digit
    __shared__ u32 lds4[WAVES_PER_BLOCK];

    const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const u32 tile_base = blockIdx.x * RS_TILE;
    const u32 tile_count = (n - tile_base) < (u32)RS_TILE ? (n - tile_base) : (u32)RS_TILE;

#pragma unroll
    for (int k = 0; k < WAVES_PER_BLOCK; k++) wave_cnt[k][tid] = 0;
    __syncthreads();

    K key[RS_IPT];
    u32 val[RS_IPT];
    u32 rank[RS_IPT];          // (rank among equal digits inside this wave's 1024 pairs)
    const u64 lt_mask = (1ull << lane) - 1ull;

#pragma unroll
    for (int j = 0; j < RS_IPT; j++) {
        const u32 local = w * RS_WAVE_ITEMS + j * WAVE + lane;
        const bool valid = local < tile_count;
        key[j] = valid ? keys_in[tile_base + local] : (K)0;
        val[j] = valid ? vals_in[tile_base + local] : 0u;
        const u32 digit = (u32)(key[j] >> shift) & 255u;
        // wave64 multisplit: lanes holding the same digit find each other with 8 ballots
        u64 mask = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const bool b = (digit >> bit) & 1u;
            const u64 bal = __ballot(b);
            mask &= b ? bal : ~bal;
        }
        if (!valid) mask = 1ull << lane;
        const u32 cnt = (u32)__popcll(mask);
        const u32 before = (u32)__popcll(mask & lt_mask);
        const int leader = __ffsll((unsigned long long)mask) - 1;
        u32 prior = 0;
        if (valid && (int)lane == leader) prior = atomicAdd(&wave_cnt[w][digit], cnt);
        prior = __shfl(prior, leader, WAVE);
        rank[j] = prior + before;
    }
    __syncthreads();

    // thread d owns digit d: wave bases, then the exclusive scan over digits inside the tile
    {
        const u32 c0 = wave_cnt[0][tid], c1 = wave_cnt[1][tid], c2 = wave_cnt[2][tid],
                  c3 = wave_cnt[3][tid];
        wave_cnt[0][tid] = 0;
        wave_cnt[1][tid] = c0;
        wave_cnt[2][tid] = c0 + c1;
        wave_cnt[3][tid] = c0 + c1 + c2;
        u32 total;
        const u32 start = block_exclusive_sum(c0 + c1 + c2 + c3, lds4, total);
        digit_start[tid] = start;
        global_base[tid] = scanned_hist[tid * n_tiles + blockIdx.x] - start;
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < RS_IPT; j++) {
        const u32 local = w * RS_WAVE_ITEMS + j * WAVE + lane;
        if (local < tile_count) {
            const u32 digit = (u32)(key[j] >> shift) & 255u;
            const u32 slot = digit_start[digit] + wave_cnt[w][digit] + rank[j];
            s_keys[slot] = key[j];
            s_vals[slot] = val[j];
        }
    }
    __syncthreads();

#pragma unroll 4
    for (int j = 0; j < RS_IPT; j++) {
        const u32 slot = j * BLOCK + tid;
        if (slot < tile_count) {
            const K k = s_keys[slot];
            const u32 dst = global_base[(u32)(k >> shift) & 255u] + slot;
            keys_out[dst] = k;
            vals_out[dst] = s_vals[slot];
        }
    }
}

template <class K> struct SortBufs {
    K *keys[2];
    u32 *vals[2];
};

// Sorts on key bits [0, bits).  Input in buffers [0]; returns the index (0/1)
// of the buffers that hold the sorted pairs.
template <class K>
static int radix_sort_pairs(Ctx &ctx, SortBufs<K> &b, u32 n, int bits)
{
    if (n == 0) return 0;
    const u32 n_tiles = ceil_div_u32(n, RS_TILE);
    const size_t mark = ctx.arena->mark();
    u32 *hist = ctx.arena->alloc<u32>((size_t)RS_BINS * n_tiles);
    int cur = 0;
    for (int shift = 0; shift < bits; shift += 8) {
        LAUNCH_NAMED(ctx, sizeof(K) == 8 ? "radix_hist_kernel<u64>" : "radix_hist_kernel<u32>",
                     (radix_hist_kernel<K>), n_tiles, (const K *)b.keys[cur], n, shift, hist, n_tiles);
        device_scan<ArrIn, false>(ctx, ArrIn{hist}, RS_BINS * n_tiles, hist);
        LAUNCH_NAMED(ctx, sizeof(K) == 8 ? "radix_scatter_kernel<u64>" : "radix_scatter_kernel<u32>",
                     (radix_scatter_kernel<K>), n_tiles, (const K *)b.keys[cur],
               (const u32 *)b.vals[cur], b.keys[cur ^ 1], b.vals[cur ^ 1], n, shift,
               (const u32 *)hist, n_tiles);
        cur ^= 1;
        if (ctx.stats) {
            ctx.stats->radix_passes++;
            ctx.stats->radix_elems += n;
            if (sizeof(K) == 8) { ctx.stats->radix_elems_u64 += n; ctx.stats->radix_passes_u64++; }
            else { ctx.stats->radix_elems_u32 += n; ctx.stats->radix_passes_u32++; }
            if ((i64)(sizeof(K) + 4) > ctx.stats->radix_elem_bytes)
                ctx.stats->radix_elem_bytes = sizeof(K) + 4;
        }
    }
    ctx.arena->release(mark);
    return cur;
}
