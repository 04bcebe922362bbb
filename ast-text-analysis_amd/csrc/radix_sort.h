// radix_sort.h -- stable LSD radix sort of (key, value) pairs, 8-bit digits.
//
// One pass:
//   radix_hist_kernel     per-tile 256-bin histogram (LDS atomics)      -> hist[tile][digit]
//   hist_chunk_sums / hist_chunk_scan / hist_apply
//                         exclusive scan in digit-major ORDER over the tile-major
//                         LAYOUT: one thread per digit column, rows of 256
//                         counters are read and written coalesced       -> global bases
//   radix_scatter_kernel  wave64 ballot multisplit for the stable rank inside
//                         the tile, tile re-ordered by digit in LDS, then
//                         written out as contiguous per-digit runs.
//
// Tile = 4096 pairs per workgroup (see radix_scatter_kernel for the geometry).
#pragma once
#include "common.h"
#include "scan.h"

#ifndef RS_THREADS
#define RS_THREADS 1024                // threads per scatter workgroup (16 waves)
#endif
#define RS_TILE 4096                   // pairs per workgroup, both kernels
#define RS_BINS 256

// Where a pass reads its pairs from: the buffers of the previous pass, or -- first pass only -- a
// generator that computes pair i on the fly (the keys then never make a round trip through HBM
// before the first scatter).
template <class K> struct PairSrc {
    const K *keys;
    const u32 *vals;
    __device__ __forceinline__ K key(u32 i) const { return keys[i]; }
    __device__ __forceinline__ u32 val(u32 i) const { return vals[i]; }
};

// digit of pair i at `shift`; a generator that can produce the first pass's digit without the whole
// key (low_digit, shift == 0 there by construction) is asked for just that
template <class Src>
__device__ __forceinline__ auto radix_digit(const Src &src, u32 i, int shift, int) -> decltype(src.low_digit(i))
{
    return src.low_digit(i);
}
template <class Src>
__device__ __forceinline__ u32 radix_digit(const Src &src, u32 i, int shift, long)
{
    return (u32)(src.key(i) >> shift) & 255u;
}

// One more key for the wave's sub-histogram.  Keys that arrive sorted on their high bits (the group
// numbers of the refinement rounds) put a whole wavefront on ONE bin, where 64 LDS atomics would queue
// up: that case is a compare against the first active lane and a single add.
__device__ __forceinline__ void radix_hist_add(u32 *mine, u32 digit)
{
    const u32 first = __builtin_amdgcn_readfirstlane(digit);
    const u64 active = __ballot(1);
    if (__ballot(digit == first) == active) {
        if (digit == first && lane_id() == (u32)(__ffsll((unsigned long long)active) - 1))
            atomicAdd(&mine[first], (u32)__popcll(active));
    } else {
        atomicAdd(&mine[digit], 1u);
    }
}

template <class K, class Src>
__global__ __launch_bounds__(BLOCK) void radix_hist_kernel(Src src, u32 n, int shift, u32 *__restrict__ hist,
                                                           u32 n_tiles)
{
    __shared__ u32 bins[WAVES_PER_BLOCK][RS_BINS];     // one sub-histogram per wave: a quarter of the atomic contention
#pragma unroll
    for (int k = 0; k < WAVES_PER_BLOCK; k++) bins[k][threadIdx.x] = 0;
    __syncthreads();
    u32 *mine = bins[wave_id()];
    const u32 base = blockIdx.x * RS_TILE;
    if constexpr (std::is_same<Src, PairSrc<K>>::value && sizeof(K) == 8) {
        if (base + RS_TILE <= n) {                      // a full tile of 64-bit keys: 16-byte loads, two keys each
#pragma unroll
            for (int j = 0; j < RS_TILE / (BLOCK * 2); j++) {
                const uint4 k2 = reinterpret_cast<const uint4 *>(src.keys + base)[j * BLOCK + threadIdx.x];
                const u64 ka = ((u64)k2.y << 32) | k2.x, kb = ((u64)k2.w << 32) | k2.z;
                radix_hist_add(mine, (u32)(ka >> shift) & 255u);
                radix_hist_add(mine, (u32)(kb >> shift) & 255u);
            }
            __syncthreads();
            hist[(size_t)blockIdx.x * RS_BINS + threadIdx.x] =
                bins[0][threadIdx.x] + bins[1][threadIdx.x] + bins[2][threadIdx.x] + bins[3][threadIdx.x];
            return;
        }
    }
    if constexpr (std::is_same<Src, PairSrc<K>>::value && sizeof(K) == 4) {
        if (base + RS_TILE <= n) {                      // a full tile of 32-bit keys: 16-byte loads
#pragma unroll
            for (int j = 0; j < RS_TILE / (BLOCK * 4); j++) {
                const uint4 k4 = reinterpret_cast<const uint4 *>(src.keys + base)[j * BLOCK + threadIdx.x];
                radix_hist_add(mine, (k4.x >> shift) & 255u);
                radix_hist_add(mine, (k4.y >> shift) & 255u);
                radix_hist_add(mine, (k4.z >> shift) & 255u);
                radix_hist_add(mine, (k4.w >> shift) & 255u);
            }
            __syncthreads();
            hist[(size_t)blockIdx.x * RS_BINS + threadIdx.x] =
                bins[0][threadIdx.x] + bins[1][threadIdx.x] + bins[2][threadIdx.x] + bins[3][threadIdx.x];
            return;
        }
    }
#pragma unroll 4
    for (int j = 0; j < RS_TILE / BLOCK; j++) {
        const u32 i = base + j * BLOCK + threadIdx.x;
        if (i < n) radix_hist_add(mine, radix_digit(src, i, shift, 0));
    }
    __syncthreads();
    static_assert(WAVES_PER_BLOCK == 4, "sub-histogram sum below assumes 4 waves");
    hist[(size_t)blockIdx.x * RS_BINS + threadIdx.x] =          // tile-major: one coalesced row
        bins[0][threadIdx.x] + bins[1][threadIdx.x] + bins[2][threadIdx.x] + bins[3][threadIdx.x];
}

// ---- scan of the histogram: base[t][d] = sum_{d' < d, all t'} h[t'][d'] + sum_{t' < t} h[t'][d] ----
#define HS_CHUNK 64                    // tiles per chunk

__global__ __launch_bounds__(BLOCK) void hist_chunk_sums_kernel(const u32 *__restrict__ hist, u32 n_tiles,
                                                                u32 *__restrict__ chunk_sums)
{
    const u32 d = threadIdx.x, t0 = blockIdx.x * HS_CHUNK;
    const u32 t1 = t0 + HS_CHUNK < n_tiles ? t0 + HS_CHUNK : n_tiles;
    u32 sum = 0;
    for (u32 t = t0; t < t1; t++) sum += hist[(size_t)t * RS_BINS + d];
    chunk_sums[(size_t)blockIdx.x * RS_BINS + d] = sum;
}

// one workgroup of 1024 threads: per digit column, exclusive scan over the chunks -- the column is cut into
// HSC_PARTS stretches scanned side by side (a single thread per column was 15 us of pure latency per pass) --
// then the digit bases
#define HSC_PARTS 4
__global__ __launch_bounds__(BLOCK * HSC_PARTS) void hist_chunk_scan_kernel(const u32 *__restrict__ chunk_sums,
                                                                            u32 n_chunks, u32 *__restrict__ chunk_prefix)
{
    __shared__ u32 part_total[HSC_PARTS][RS_BINS];
    __shared__ u32 lds4[WAVES_PER_BLOCK];
    const u32 d = threadIdx.x & (RS_BINS - 1u), g = threadIdx.x / RS_BINS;
    const u32 per = (n_chunks + HSC_PARTS - 1u) / HSC_PARTS;
    const u32 c0 = g * per < n_chunks ? g * per : n_chunks;
    const u32 c1 = c0 + per < n_chunks ? c0 + per : n_chunks;
    u32 run = 0;
#pragma unroll 8
    for (u32 c = c0; c < c1; c++) run += chunk_sums[(size_t)c * RS_BINS + d];
    part_total[g][d] = run;
    __syncthreads();
    u32 before = 0, column = 0;                 // chunks of this column in earlier stretches / in all stretches
#pragma unroll
    for (u32 k = 0; k < HSC_PARTS; k++) {
        const u32 v = part_total[k][d];
        if (k < g) before += v;
        column += v;
    }
    run = before;
#pragma unroll 8
    for (u32 c = c0; c < c1; c++) {             // separate in/out arrays: the loads pipeline
        const u32 v = chunk_sums[(size_t)c * RS_BINS + d];
        chunk_prefix[(size_t)c * RS_BINS + d] = run;
        run += v;
    }
    if (g == 0) {                               // (the first 256 threads = 4 waves: the block-wide sum helper fits)
        u32 total;
        const u32 digit_base = block_exclusive_sum(column, lds4, total);     // all smaller digits, all tiles
        chunk_prefix[(size_t)n_chunks * RS_BINS + d] = digit_base;
    }
}

__global__ __launch_bounds__(BLOCK) void hist_apply_kernel(u32 *__restrict__ hist, u32 n_tiles,
                                                           const u32 *__restrict__ chunk_sums, u32 n_chunks)
{
    const u32 d = threadIdx.x, t0 = blockIdx.x * HS_CHUNK;
    const u32 t1 = t0 + HS_CHUNK < n_tiles ? t0 + HS_CHUNK : n_tiles;
    u32 run = chunk_sums[(size_t)n_chunks * RS_BINS + d] + chunk_sums[(size_t)blockIdx.x * RS_BINS + d];
    for (u32 t = t0; t < t1; t++) {
        const u32 v = hist[(size_t)t * RS_BINS + d];
        hist[(size_t)t * RS_BINS + d] = run;
        run += v;
    }
}

// Few tiles (small sorts are bound by launch latency): the whole scan in one workgroup, one launch
// instead of three.  Thread d owns digit column d: exclusive scan down the tiles, then the digit bases.
#define HS_SMALL_TILES 128
__global__ __launch_bounds__(BLOCK) void hist_scan_small_kernel(u32 *__restrict__ hist, u32 n_tiles)
{
    __shared__ u32 lds4[WAVES_PER_BLOCK];
    const u32 d = threadIdx.x;
    u32 run = 0;
#pragma unroll 8
    for (u32 t = 0; t < n_tiles; t++) run += hist[(size_t)t * RS_BINS + d];
    u32 total;
    u32 base = block_exclusive_sum(run, lds4, total);      // all smaller digits, all tiles
#pragma unroll 8
    for (u32 t = 0; t < n_tiles; t++) {
        const u32 v = hist[(size_t)t * RS_BINS + d];
        hist[(size_t)t * RS_BINS + d] = base;
        base += v;
    }
}

// THREADS = 64*WAVES threads move one 4096-pair tile.  Wave w owns the contiguous
// 4096/WAVES pairs [w*64*IPT, (w+1)*64*IPT) and walks them in IPT rows of 64
// (coalesced), so (wave, row, lane) is the input order and the pass is stable.
// Keys and values are staged through the SAME LDS buffer one after the other
// (the per-pair destination is kept in registers), which keeps the workgroup at
// 4096*sizeof(K) + 4*WAVES*256 + 2 KiB of LDS: two 16-wave workgroups (32 waves,
// the hardware maximum) fit a CU.
template <class K, int WAVES> struct ScatterLds {      // (declared once in the kernel: both tile paths share it)
    K s_keys[RS_TILE];
    u32 wave_cnt[WAVES][RS_BINS];           // per-wave digit counts, then wave bases
    u32 digit_start[RS_BINS];               // first slot of the digit inside the tile
    u32 global_base[RS_BINS];               // output index of slot 0 of the digit
    u32 lds4[4];
};

template <class K, int THREADS, class Src, bool FULL>
__device__ __forceinline__ void radix_scatter_tile(
    ScatterLds<K, THREADS / WAVE> &lds, const Src &src, K *__restrict__ keys_out, u32 *__restrict__ vals_out, int shift,
    const u32 *__restrict__ scanned_hist, u32 tile, u32 tile_count)
{
    constexpr int WAVES = THREADS / WAVE;
    constexpr int IPT = RS_TILE / THREADS;
    constexpr int WAVE_ITEMS = WAVE * IPT;
    static_assert(THREADS >= RS_BINS && RS_TILE % THREADS == 0, "bad scatter geometry");
    K (&s_keys)[RS_TILE] = lds.s_keys;
    u32 (&wave_cnt)[WAVES][RS_BINS] = lds.wave_cnt;
    u32 (&digit_start)[RS_BINS] = lds.digit_start;
    u32 (&global_base)[RS_BINS] = lds.global_base;
    u32 (&lds4)[4] = lds.lds4;
    u32 *s_vals = reinterpret_cast<u32 *>(s_keys);

    const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const u32 tile_base = tile * RS_TILE;

    for (u32 i = tid; i < WAVES * RS_BINS; i += THREADS) (&wave_cnt[0][0])[i] = 0;
    __syncthreads();

    K key[IPT];
    u32 val[IPT];
    u32 slot[IPT];             // rank among equal digits inside this wave's pairs, later the tile slot

#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const u32 local = w * WAVE_ITEMS + j * WAVE + lane;
        const bool valid = FULL || local < tile_count;
        key[j] = valid ? src.key(tile_base + local) : (K)0;
        val[j] = valid ? src.val(tile_base + local) : 0u;
        const u32 digit = (u32)(key[j] >> shift) & 255u;
        // wave64 multisplit: lanes holding the same digit find each other with 8 ballots.  Per bit:
        // s = the bit spread over a word (one v_bfe_i32), the ballot of it, and the lanes that
        // differ from this one in that bit accumulate in `diff` (xor + or per half).  (The kernel is
        // bound by instruction issue, not by HBM: 9 -> 5 VALU per bit here was worth 10 % of a pass;
        // interleaving the rows' chains to fill the ballot hazards was measured and is slower --
        // the rows then cannot start before all four loads have landed.)
        u32 diff_lo = 0, diff_hi = 0;
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const u32 sbit = (u32)__builtin_amdgcn_sbfe((int)digit, (u32)bit, 1u);      // 0 or ~0
            const u64 bal = __ballot((int)sbit < 0);
            diff_lo |= (u32)bal ^ sbit;
            diff_hi |= (u32)(bal >> 32) ^ sbit;
        }
        u64 mask = ~(((u64)diff_hi << 32) | diff_lo);
        if (!FULL) {
            mask &= __ballot(valid);
            if (!valid) mask = 1ull << lane;
        }
        const u32 cnt = (u32)__popcll(mask);
        // set bits of `mask` below this lane (v_mbcnt_lo/hi take a per-lane mask)
        const u32 before = __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
        const int leader = __ffsll((unsigned long long)mask) - 1;
        u32 prior = 0;
        if (valid && before == 0) prior = atomicAdd(&wave_cnt[w][digit], cnt);
        prior = __shfl(prior, leader, WAVE);
        slot[j] = prior + before;
    }
    __syncthreads();

    // thread d (< 256) owns digit d: wave bases, then the exclusive scan over digits inside the tile
    u32 total_d = 0, inc = 0;
    if (tid < RS_BINS) {
        u32 run = 0;
#pragma unroll
        for (int k = 0; k < WAVES; k++) {
            const u32 c = wave_cnt[k][tid];
            wave_cnt[k][tid] = run;
            run += c;
        }
        total_d = run;
        inc = wave_inclusive_sum(total_d);
        if (lane == 63) lds4[w] = inc;
    }
    __syncthreads();
    if (tid < RS_BINS) {
        u32 start = inc - total_d;
        if (w > 0) start += lds4[0];
        if (w > 1) start += lds4[1];
        if (w > 2) start += lds4[2];
        digit_start[tid] = start;
        global_base[tid] = scanned_hist[(size_t)tile * RS_BINS + tid] - start;
    }
    __syncthreads();

    // keys: scatter into tile order, then stream out as contiguous per-digit runs
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const u32 local = w * WAVE_ITEMS + j * WAVE + lane;
        if (FULL || local < tile_count) {
            const u32 digit = (u32)(key[j] >> shift) & 255u;
            slot[j] = digit_start[digit] + wave_cnt[w][digit] + slot[j];
            s_keys[slot[j]] = key[j];
        }
    }
    __syncthreads();
    u32 dst[IPT];
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const u32 pos = j * THREADS + tid;
        if (FULL || pos < tile_count) {
            const K k = s_keys[pos];
            dst[j] = global_base[(u32)(k >> shift) & 255u] + pos;
            keys_out[dst[j]] = k;
        }
    }
    __syncthreads();
    // values: same slots, same destinations, through the same LDS buffer
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const u32 local = w * WAVE_ITEMS + j * WAVE + lane;
        if (FULL || local < tile_count) s_vals[slot[j]] = val[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const u32 pos = j * THREADS + tid;
        if (FULL || pos < tile_count) vals_out[dst[j]] = s_vals[pos];
    }
}

template <class K, int THREADS, class Src>
__global__ __launch_bounds__(THREADS) void radix_scatter_kernel(
    Src src, K *__restrict__ keys_out, u32 *__restrict__ vals_out, u32 n, int shift,
    const u32 *__restrict__ scanned_hist, u32 n_tiles)
{
    // XCD-aware tile order: workgroups go round-robin over the 8 XCDs, each with its own L2.  The
    // per-digit runs of NEIGHBOURING tiles are neighbours in the output, so neighbouring tiles are
    // given to the same XCD (XCD x takes the x-th eighth of the tiles): the partial lines at the
    // seams of the runs meet in one L2 instead of reaching memory as masked writes from two.
    const u32 per_xcd = (n_tiles + 7u) / 8u;
    const u32 tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const u32 tile_count = (n - tile * RS_TILE) < (u32)RS_TILE ? (n - tile * RS_TILE) : (u32)RS_TILE;
    // every tile but the last is full: no bounds tests in its code path
    __shared__ ScatterLds<K, THREADS / WAVE> lds;
    if (tile_count == (u32)RS_TILE)
        radix_scatter_tile<K, THREADS, Src, true>(lds, src, keys_out, vals_out, shift, scanned_hist, tile, tile_count);
    else
        radix_scatter_tile<K, THREADS, Src, false>(lds, src, keys_out, vals_out, shift, scanned_hist, tile, tile_count);
}

template <class K> struct SortBufs {
    K *keys[2];
    u32 *vals[2];
};

struct NoGen {};

// Sorts on key bits [begin_bit, bits).  Input in buffers [0] -- or, with a generator (a Src with
// key(i) / val(i)), produced on the fly by the first pass, which then writes into buffers [0].
// Returns the index (0/1) of the buffers that hold the sorted pairs.
template <class K, class Gen = NoGen>
static int radix_sort_pairs(Ctx &ctx, SortBufs<K> &b, u32 n, int bits, int begin_bit = 0, Gen gen = Gen())
{
    constexpr bool HAS_GEN = !std::is_same<Gen, NoGen>::value;
    if (n == 0) return 0;
    const u32 n_tiles = ceil_div_u32(n, RS_TILE);
    const size_t mark = ctx.arena->mark();
    u32 *hist = ctx.arena->alloc<u32>((size_t)RS_BINS * n_tiles);
    const u32 n_chunks = ceil_div_u32(n_tiles, HS_CHUNK);
    u32 *chunk_sums = ctx.arena->alloc<u32>((size_t)RS_BINS * n_chunks);
    u32 *chunk_prefix = ctx.arena->alloc<u32>((size_t)RS_BINS * (n_chunks + 1));
    const bool prof = ctx.prof && ctx.prof->enabled;
    int cur = 0;
    bool first = HAS_GEN;                       // the generator pass reads no buffers and writes [0]
    for (int shift = begin_bit; shift < bits; shift += 8) {
        const PairSrc<K> src{b.keys[cur], b.vals[cur]};
        const int out = first ? 0 : cur ^ 1;
        if constexpr (HAS_GEN) {
            if (first)
                LAUNCH_NAMED(ctx, sizeof(K) == 8 ? "radix_hist_kernel<u64,gen>" : "radix_hist_kernel<u32,gen>",
                             (radix_hist_kernel<K, Gen>), n_tiles, gen, n, shift, hist, n_tiles);
        }
        if (!first)
            LAUNCH_NAMED(ctx, sizeof(K) == 8 ? "radix_hist_kernel<u64>" : "radix_hist_kernel<u32>",
                         (radix_hist_kernel<K, PairSrc<K>>), n_tiles, src, n, shift, hist, n_tiles);
        if (n_tiles <= HS_SMALL_TILES) {
            LAUNCH(ctx, hist_scan_small_kernel, 1, hist, n_tiles);
        } else {
            LAUNCH(ctx, hist_chunk_sums_kernel, n_chunks, (const u32 *)hist, n_tiles, chunk_sums);
            LAUNCH_BLOCK(ctx, hist_chunk_scan_kernel, 1, BLOCK * HSC_PARTS, (const u32 *)chunk_sums, n_chunks, chunk_prefix);
            LAUNCH(ctx, hist_apply_kernel, n_chunks, hist, n_tiles, (const u32 *)chunk_prefix, n_chunks);
        }
        if (!ctx.dry) {
            if constexpr (HAS_GEN) {
                if (first) {
                    if (prof) ctx.prof->begin(sizeof(K) == 8 ? "radix_scatter_kernel<u64,gen>" : "radix_scatter_kernel<u32,gen>", ctx.stream);
                    hipLaunchKernelGGL((radix_scatter_kernel<K, RS_THREADS, Gen>), dim3(8 * ((n_tiles + 7) / 8)), dim3(RS_THREADS), 0,
                                       ctx.stream, gen, b.keys[out], b.vals[out], n, shift, (const u32 *)hist, n_tiles);
                }
            }
            if (!first) {
                if (prof) ctx.prof->begin(sizeof(K) == 8 ? "radix_scatter_kernel<u64>" : "radix_scatter_kernel<u32>", ctx.stream);
                hipLaunchKernelGGL((radix_scatter_kernel<K, RS_THREADS, PairSrc<K>>), dim3(8 * ((n_tiles + 7) / 8)), dim3(RS_THREADS), 0,
                                   ctx.stream, src, b.keys[out], b.vals[out], n, shift, (const u32 *)hist, n_tiles);
            }
            HIP_CHECK(hipGetLastError());
            if (prof) ctx.prof->end(ctx.stream);
        }
        cur = out;
        if (ctx.stats) {
            ctx.stats->radix_passes++;
            if (!first) {                       // (generator passes are accounted under their own kernel names)
                ctx.stats->radix_elems += n;
                if (sizeof(K) == 8) { ctx.stats->radix_elems_u64 += n; ctx.stats->radix_passes_u64++; }
                else { ctx.stats->radix_elems_u32 += n; ctx.stats->radix_passes_u32++; }
            }
            if ((i64)(sizeof(K) + 4) > ctx.stats->radix_elem_bytes)
                ctx.stats->radix_elem_bytes = sizeof(K) + 4;
        }
        first = false;
    }
    ctx.arena->release(mark);
    return cur;
}
