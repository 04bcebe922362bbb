// radix_sort.h -- stable LSD radix sort of (key, value) pairs, 8-bit digits.
//
// One pass = three launches (no inter-workgroup communication inside a launch):
//   radix_hist_kernel     a workgroup counts a GROUP of RS_GROUP consecutive tiles, every wave whole tiles
//                         into per-tile histograms in LDS, and writes them as running sums -- hist[tile][digit]
//                         = the tile's exclusive prefix inside its group -- plus the group's sums; the digit
//                         totals over all groups are accumulated on the way (global atomics, one row per group)
//   radix_spine_kernel    exclusive scan down the groups for every digit column + the digit bases
//   radix_scatter_kernel  one tile per workgroup: wave64 ballot multisplit for the stable rank inside the
//                         tile, tile re-ordered by digit in LDS, written out as contiguous per-digit runs
//                         at  group_prefix[group][digit] + hist[tile][digit].
// (Round 1 scanned the per-tile histograms with three more streaming launches per pass; the group
// structure leaves a scan over 1/8 of the rows.  Wider digits -- 10 bits, 3 passes for a 30-bit key --
// were built and measured: the pass is bound by the number of write requests a CU can keep in flight
// towards the L2, a request carries one 64-byte-aligned piece of a run, and runs of 8 pairs make a
// 3-pass sort as slow as the 4-pass one; see DESIGN.md.)
#pragma once
#include "common.h"
#include "scan.h"

#define RS_THREADS 1024                // threads per scatter workgroup (16 waves)
#ifndef RS_IPT
#define RS_IPT 4
#endif
#define RS_TILE (RS_THREADS * RS_IPT)  // 4096 pairs per workgroup
#define RS_DB 8
#define RS_BINS 256
#ifndef RS_GROUP
#define RS_GROUP 8                     // tiles per histogram group
#endif
#define RS_HIST_THREADS 256
#ifndef RS_TOTAL_SHARDS
#define RS_TOTAL_SHARDS 8
#endif

// Segmented sort (RsSeg): the n pairs are the suffixes of several DOCUMENTS laid side by side -- document d owns the
// positions [doc_off[d], doc_off[d + 1]) of the input AND of the output --, and every pass is a stable counting sort of
// each document on its own: the document number never has to be a key digit (for 256 documents that is a whole pass
// over HBM saved, or -- what window_sort.h does with it -- 8 more key bits of text in the same passes).  The tiles are
// cut per document (the last one of a document is short) and so are the histogram groups (the last group of a document
// may hold fewer than RS_GROUP tiles with pairs in them: "virtual" tiles, a workgroup that returns at once); a tile's
// digit base is  doc_off[d] + (the document's pairs with smaller digits) + (the digit in the document's earlier tiles),
// which is the spine's column scan run per document.  Meant for documents of a histogram group or more (window_sort.h decides); a shard of
// many small ones keeps the document number in the key.
#define RS_SEG_MAX_DOCS 65535              // (the spine's grid: one row of workgroups per document)
struct RsSeg {
    const u32 *group_doc = nullptr;     // the document of (virtual) group g
    const u32 *doc_group0 = nullptr;    // the first group of document d (n_docs + 1 entries)
    const u32 *doc_off = nullptr;       // n_docs + 1 offsets
    u32 n_docs = 0;                     // 0: one segment, tiles of RS_TILE pairs from position 0 on
    u32 n_groups = 0;                   // all documents' groups
    u32 shards = 1;                     // copies of a document's digit totals (see RS_TOTAL_SHARDS)
    // the spine: documents of at most RS_SPINE_DOC_GROUPS groups take radix_spine_docs_kernel (one workgroup each), the
    // n_big larger ones -- listed in big_docs -- the column-parallel kernel
    const u32 *big_docs = nullptr;
    u32 n_big = 0;
};
#define RS_SPINE_DOC_GROUPS 64          // (2 M pairs)
// the pairs of (virtual) tile `tile`: [base, base + count)
__device__ __forceinline__ void rs_tile_range(const RsSeg &seg, u32 n, u32 tile, u32 &base, u32 &count)
{
    if (!seg.n_docs) {
        base = tile * (u32)RS_TILE;
        count = n - base < (u32)RS_TILE ? n - base : (u32)RS_TILE;
        return;
    }
    const u32 d = seg.group_doc[tile / RS_GROUP];
    const u32 t_in = tile - seg.doc_group0[d] * RS_GROUP;
    const u32 end = seg.doc_off[d + 1];
    base = seg.doc_off[d] + t_in * (u32)RS_TILE;        // (< n + RS_GROUP * RS_TILE < 2^32)
    count = base < end ? (end - base < (u32)RS_TILE ? end - base : (u32)RS_TILE) : 0u;
}

// Where a pass reads its pairs from: the buffers of the previous pass, or -- first pass only -- a
// generator (MODE 1: computes pair i on the fly, key(i) / val(i); MODE 2: fills a whole tile of keys in
// LDS, see TextWindowGen in window_sort.h; the value of element i is i).  The keys then never make a
// round trip through HBM before the first scatter.
template <class K> struct PairSrc {
    static constexpr int MODE = 0;
    const K *keys;
    const u32 *vals;
    __device__ __forceinline__ K key(u32 i) const { return keys[i]; }
    __device__ __forceinline__ u32 val(u32 i) const { return vals[i]; }
};

// One more key for the wave's sub-histogram.  Keys that arrive sorted on their high bits (the group
// numbers of the refinement rounds) put a whole wavefront on ONE bin, where 64 LDS atomics would queue
// up: that case is a compare against the first active lane and a single add.
template <bool CHECK = true> __device__ __forceinline__ void radix_hist_add(u32 *mine, u32 digit)
{
    // (CHECK = false: the caller knows the digits of a wavefront are spread out -- the low digits of window keys in text
    // order --, and the test for the one-bin case, two ballots per key, is saved)
    if constexpr (!CHECK) { atomicAdd(&mine[digit], 1u); return; }
    const u32 first = __builtin_amdgcn_readfirstlane(digit);
    const u64 active = __ballot(1);
    if (__ballot(digit == first) == active) {
        if (digit == first && lane_id() == (u32)(__ffsll((unsigned long long)active) - 1))
            atomicAdd(&mine[first], (u32)__popcll(active));
    } else {
        atomicAdd(&mine[digit], 1u);
    }
}

// ---- histogram: per-tile counts of a group of tiles, written as running sums ------------------------------
// Wave w of the workgroup counts the tiles w, w + 4, ... of its group into that tile's OWN histogram in
// LDS: no wave ever touches another wave's counters, so there is no barrier inside the loop, and a wave
// keeps a batch of 16-byte loads in flight while it counts the batch before.  At the end thread d turns
// digit d's column into running sums: hist[tile][d] = the pairs with digit d in the group's earlier tiles.
#ifndef RS_HIST_BATCH
#define RS_HIST_BATCH 8                // 16-byte loads a wave keeps in flight (8 KiB)
#endif
// Copies of a tile's histogram, one per lane class (lane mod RS_HIST_COPIES): natural-language text puts several lanes
// of a wavefront on the same few bins, and LDS atomics to one address are served one after the other -- the Zipf
// stand-in counted at 2.0 TB/s where uniform text reaches 3.7.  Lanes of different classes never meet in a counter.
// (4 copies when the document number sat in the keys; with text only in them -- the segmented sort -- 2 are as good on the
// Zipf stand-in, 0.375 against 0.39 ms, and better on uniform text, where the copies cost occupancy: configs[2] 0.49 + 0.20
// against 0.51 + 0.24 ms for the two histogram kernels, 1 copy 0.53 + 0.20 and 0.585 on the Zipf stand-in)
#ifndef RS_HIST_COPIES
#define RS_HIST_COPIES 2
#endif
template <class K, class Src, bool CHECK = true>
__global__ __launch_bounds__(RS_HIST_THREADS) void radix_hist_kernel(Src src, u32 n, int shift, u32 mask, u32 n_tiles,
                                                                     u32 *__restrict__ hist, u32 *__restrict__ group_sum,
                                                                     u32 *__restrict__ digit_total, RsSeg seg)
{
    constexpr int UW = RS_HIST_THREADS / WAVE;
    static_assert(RS_HIST_THREADS == RS_BINS, "thread d owns digit d");
    __shared__ u32 bins[RS_GROUP][RS_HIST_COPIES][RS_BINS];
    if constexpr (Src::MODE == 2) src.prepare();
#pragma unroll
    for (int k = 0; k < RS_GROUP; k++)
#pragma unroll
        for (int c = 0; c < RS_HIST_COPIES; c++) bins[k][c][threadIdx.x] = 0;
    __syncthreads();
    const u32 g = blockIdx.x, lane = lane_id();
    const u32 t0 = g * RS_GROUP, t1 = t0 + RS_GROUP < n_tiles ? t0 + RS_GROUP : n_tiles;
    for (u32 tile = t0 + wave_id(); tile < t1; tile += UW) {
        u32 *mine = bins[tile - t0][lane & (RS_HIST_COPIES - 1)];
        u32 base, count;
        rs_tile_range(seg, n, tile, base, count);
        if (count == 0) continue;
        if constexpr (Src::MODE == 0) {
            constexpr u32 PER = 16 / sizeof(K);         // keys per 16-byte load
            constexpr int LOADS = RS_TILE / (WAVE * PER);
            static_assert(LOADS % RS_HIST_BATCH == 0, "batches of loads");
            if (count == (u32)RS_TILE) {
                // (a document's tiles start anywhere: 16-byte loads at 4-byte alignment)
                const char *p = reinterpret_cast<const char *>(src.keys + base) + 16u * lane;
                for (int j0 = 0; j0 < LOADS; j0 += RS_HIST_BATCH) {
                    uint4 q[RS_HIST_BATCH];
#pragma unroll
                    for (int j = 0; j < RS_HIST_BATCH; j++) __builtin_memcpy(&q[j], p + (size_t)(j0 + j) * WAVE * 16u, 16);
                    // (nothing moves across this line: without the one-bin test the loop body is one basic block, and the
                    // scheduler -- after the fewest registers -- put every load next to its four adds, each with a wait of
                    // its own: one load in flight per wave instead of the batch)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < RS_HIST_BATCH; j++) {
                        if constexpr (sizeof(K) == 8) {
                            const u64 ka = ((u64)q[j].y << 32) | q[j].x, kb = ((u64)q[j].w << 32) | q[j].z;
                            radix_hist_add<CHECK>(mine, (u32)(ka >> shift) & mask);
                            radix_hist_add<CHECK>(mine, (u32)(kb >> shift) & mask);
                        } else {
                            radix_hist_add<CHECK>(mine, (u32)(q[j].x >> shift) & mask);
                            radix_hist_add<CHECK>(mine, (u32)(q[j].y >> shift) & mask);
                            radix_hist_add<CHECK>(mine, (u32)(q[j].z >> shift) & mask);
                            radix_hist_add<CHECK>(mine, (u32)(q[j].w >> shift) & mask);
                        }
                    }
                }
            } else {
                for (u32 i = lane; i < count; i += WAVE) radix_hist_add<CHECK>(mine, (u32)(src.keys[base + i] >> shift) & mask);
            }
        } else if constexpr (Src::MODE == 1) {
            for (u32 i = lane; i < count; i += WAVE) radix_hist_add<CHECK>(mine, (u32)(src.key(base + i) >> shift) & mask);
        } else {
            src.template hist_tile<CHECK>(mine, base, base + count, shift, mask);
        }
    }
    syncthreads_after_lds_atomics();
    u32 run = 0;
#pragma unroll
    for (int k = 0; k < RS_GROUP; k++) {
        if (t0 + k < t1) hist[(size_t)(t0 + k) * RS_BINS + threadIdx.x] = run;
#pragma unroll
        for (int c = 0; c < RS_HIST_COPIES; c++) run += bins[k][c][threadIdx.x];
    }
    group_sum[(size_t)g * RS_BINS + threadIdx.x] = run;     // one coalesced row per group
    // (RS_TOTAL_SHARDS copies of the totals: thousands of workgroups adding into one 1 KiB row queue up behind one another)
    // (segmented: a row of totals per document)
    const u32 row = seg.n_docs ? seg.group_doc[g] * seg.shards + g % seg.shards : g % RS_TOTAL_SHARDS;
    if (run) atomicAdd(&digit_total[(size_t)row * RS_BINS + threadIdx.x], run);
}

// ---- spine: exclusive scan down the groups, per digit column, plus the digit bases --------------------
// A workgroup owns 8 digit columns; its 256 threads cut the column into 32 stretches scanned side by
// side (all of it lives in the L2: the rows were written by the kernel before).  digit_total is read
// by every workgroup (its exclusive scan = the digit bases); workgroup 0 clears the OTHER total buffer
// for the next pass's histogram.
#ifndef RS_SPINE_COLS
#define RS_SPINE_COLS 8
#endif
#define RS_SPINE_PARTS (BLOCK / RS_SPINE_COLS)
#define RS_SPINE_HELD 64u
__global__ __launch_bounds__(BLOCK) void radix_spine_kernel(const u32 *__restrict__ group_sum, u32 n_groups,
                                                            const u32 *__restrict__ digit_total,
                                                            u32 *__restrict__ next_total, u32 *__restrict__ group_prefix,
                                                            RsSeg seg)
{
    static_assert(RS_BINS == BLOCK, "one digit per thread in the scan of the totals");
    __shared__ u32 base_of[RS_BINS];
    __shared__ u32 part_total[RS_SPINE_PARTS][RS_SPINE_COLS];
    __shared__ u32 lds4[WAVES_PER_BLOCK];
    // (segmented: blockIdx.y = the document, or its place in the list of large documents -- its groups, its row of totals,
    // its first output position)
    const u32 doc = seg.big_docs ? seg.big_docs[blockIdx.y] : blockIdx.y;
    const u32 g_first = seg.n_docs ? seg.doc_group0[doc] : 0u;
    const u32 g_count = seg.n_docs ? seg.doc_group0[doc + 1] - g_first : n_groups;
    const u32 shards = seg.n_docs ? seg.shards : (u32)RS_TOTAL_SHARDS;
    {
        const size_t row0 = (size_t)doc * shards * RS_BINS;
        u32 t = 0;
        for (u32 k = 0; k < shards; k++) t += digit_total[row0 + k * RS_BINS + threadIdx.x];
        u32 total;
        base_of[threadIdx.x] = block_exclusive_sum(t, lds4, total) + (seg.n_docs ? seg.doc_off[doc] : 0u);
        if (blockIdx.x == 0)
            for (u32 k = 0; k < shards; k++) next_total[row0 + k * RS_BINS + threadIdx.x] = 0;
    }
    const u32 c = threadIdx.x & (RS_SPINE_COLS - 1u), part = threadIdx.x / RS_SPINE_COLS;
    const u32 col = blockIdx.x * RS_SPINE_COLS + c;
    const u32 per = (g_count + RS_SPINE_PARTS - 1u) / RS_SPINE_PARTS;
    const u32 r0 = g_first + (part * per < g_count ? part * per : g_count);
    const u32 r1 = r0 + per < g_first + g_count ? r0 + per : g_first + g_count;
    u32 run = 0;
    if (per <= RS_SPINE_HELD) {
        // A stretch of at most RS_SPINE_HELD rows (up to 2 048 groups = 67 M pairs): its sums are requested all at once and
        // stay in registers for the second step -- the two loops below fetch them eight at a time, twice, sixteen round
        // trips where one does (three launches per build of 16 us each, on 32 of the 256 CUs).
        u32 x[RS_SPINE_HELD];
#pragma unroll
        for (u32 i = 0; i < RS_SPINE_HELD; i++) x[i] = group_sum[(size_t)(r0 + i < r1 ? r0 + i : 0u) * RS_BINS + col];
#pragma unroll
        for (u32 i = 0; i < RS_SPINE_HELD; i++) run += r0 + i < r1 ? x[i] : 0u;
        part_total[part][c] = run;
        __syncthreads();
        u32 before = base_of[col];
        for (u32 k = 0; k < part; k++) before += part_total[k][c];
        run = before;
#pragma unroll
        for (u32 i = 0; i < RS_SPINE_HELD; i++) {
            if (r0 + i < r1) group_prefix[(size_t)(r0 + i) * RS_BINS + col] = run;
            run += x[i];
        }
        return;
    }
#pragma unroll 8
    for (u32 r = r0; r < r1; r++) run += group_sum[(size_t)r * RS_BINS + col];
    part_total[part][c] = run;
    __syncthreads();
    u32 before = base_of[col];
    for (u32 k = 0; k < part; k++) before += part_total[k][c];
    run = before;
#pragma unroll 8
    for (u32 r = r0; r < r1; r++) {                     // separate in/out arrays: the loads pipeline
        const u32 v = group_sum[(size_t)r * RS_BINS + col];
        group_prefix[(size_t)r * RS_BINS + col] = run;
        run += v;
    }
}

// The spine of a segmented sort of MANY documents of a few groups each: one workgroup per document, thread d owns
// digit d -- the document's digit bases from its row of totals, then its groups one after the other (a row of 1 KiB
// per step, coalesced).  (The column-parallel kernel above starts 32 workgroups per document, each of which scans all
// 256 totals: 4 096 documents of 64 KiB spent 0.47 ms of a 7.7 ms build there.)
__global__ __launch_bounds__(BLOCK) void radix_spine_docs_kernel(const u32 *__restrict__ group_sum,
                                                                 const u32 *__restrict__ digit_total,
                                                                 u32 *__restrict__ next_total, u32 *__restrict__ group_prefix,
                                                                 RsSeg seg)
{
    static_assert(RS_BINS == BLOCK, "one digit per thread");
    __shared__ u32 lds4[WAVES_PER_BLOCK];
    const u32 doc = blockIdx.x;
    const u32 g0 = seg.doc_group0[doc], g1 = seg.doc_group0[doc + 1];
    if (g1 - g0 > (u32)RS_SPINE_DOC_GROUPS) return;     // (a large document: the column-parallel kernel)
    const size_t row0 = (size_t)doc * seg.shards * RS_BINS;
    // (the document's group sums -- at most RS_SPINE_DOC_GROUPS rows -- are requested with its totals, all at once: fetched
    // four at a time behind the scan of the totals, a document of 30 groups waited for nine round trips)
    u32 x[RS_SPINE_DOC_GROUPS];
#pragma unroll
    for (u32 i = 0; i < RS_SPINE_DOC_GROUPS; i++) x[i] = group_sum[(size_t)(g0 + i < g1 ? g0 + i : 0u) * RS_BINS + threadIdx.x];
    u32 t = 0;
    for (u32 k = 0; k < seg.shards; k++) {
        t += digit_total[row0 + k * RS_BINS + threadIdx.x];
        next_total[row0 + k * RS_BINS + threadIdx.x] = 0;
    }
    u32 total;
    u32 run = block_exclusive_sum(t, lds4, total) + seg.doc_off[doc];
#pragma unroll
    for (u32 i = 0; i < RS_SPINE_DOC_GROUPS; i++) {
        if (g0 + i < g1) group_prefix[(size_t)(g0 + i) * RS_BINS + threadIdx.x] = run;
        run += x[i];
    }
}

// ---- scatter -----------------------------------------------------------------------------------------
// RS_THREADS = 64*WAVES threads move one 4096-pair tile.  Wave w owns the contiguous 4096/WAVES pairs
// [w*64*IPT, (w+1)*64*IPT) and walks them in IPT rows of 64 (coalesced), so (wave, row, lane) is the
// input order and the pass is stable.  Keys and values are staged through the SAME LDS buffer one after
// the other (the per-pair destination is kept in registers), which keeps the workgroup at
// 4096*sizeof(K) + 2*WAVES*256 + 2 KiB of LDS: two 16-wave workgroups (32 waves, the hardware maximum)
// fit a CU.  The per-wave digit counters are 16 bits wide, two to a word (a wave holds 256 pairs).
template <class K> struct ScatterLds {
    alignas(16) K s_keys[RS_TILE];                  // (the key generators store whole runs with 16-byte LDS writes)
    u32 wave_cnt[RS_THREADS / WAVE][RS_BINS / 2];   // per-wave digit counts (packed pairs), then wave bases
    u32 digit_start[RS_BINS];                       // first slot of the digit inside the tile
    u32 global_base[RS_BINS];                       // output index of slot 0 of the digit
    u32 wsum[RS_THREADS / WAVE];
};

template <class K, class Src, bool FULL>
__device__ __forceinline__ void radix_scatter_tile(ScatterLds<K> &lds, const Src &src, K *__restrict__ keys_out,
                                                   u32 *__restrict__ vals_out, int shift, u32 mask,
                                                   const u32 *__restrict__ hist, const u32 *__restrict__ group_prefix,
                                                   u32 tile, u32 tile_base, u32 tile_count)
{
    constexpr int WAVES = RS_THREADS / WAVE, IPT = RS_IPT, THREADS = RS_THREADS, BINS = RS_BINS, DB = RS_DB;
    constexpr int WAVE_ITEMS = WAVE * IPT;
    u32 *s_vals = reinterpret_cast<u32 *>(lds.s_keys);
    const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();

    K key[IPT];
    u32 val[IPT];
    if constexpr (Src::MODE != 2) {
#pragma unroll
        for (int j = 0; j < IPT; j++) {
            const u32 local = w * WAVE_ITEMS + j * WAVE + lane;
            const bool valid = FULL || local < tile_count;
            key[j] = valid ? src.key(tile_base + local) : (K)0;
            val[j] = valid ? src.val(tile_base + local) : 0u;
        }
    }
    // where the tile's digits go: requested now, needed after the ranking (digit pair 2*tid, 2*tid + 1)
    uint2 out_base = {0u, 0u};
    if (tid < BINS / 2) {
        const uint2 a = reinterpret_cast<const uint2 *>(hist + (size_t)tile * BINS)[tid];
        const uint2 b = reinterpret_cast<const uint2 *>(group_prefix + (size_t)(tile / RS_GROUP) * BINS)[tid];
        out_base = uint2{a.x + b.x, a.y + b.y};
    }
    if constexpr (Src::MODE == 2) {
        // the generator fills s_keys with the tile's keys in input order
        src.prepare();
        src.fill_tile(lds.s_keys, tile_base, tile_count);
        __syncthreads();
    }
    for (u32 i = tid; i < WAVES * (BINS / 2); i += THREADS) (&lds.wave_cnt[0][0])[i] = 0;
    __syncthreads();

    u32 slot[IPT];             // rank among equal digits inside this wave's pairs, later the tile slot
    if constexpr (Src::MODE == 2) {
#pragma unroll
        for (int j = 0; j < IPT; j++) {
            const u32 local = w * WAVE_ITEMS + j * WAVE + lane;
            key[j] = lds.s_keys[local];
            val[j] = tile_base + local;
        }
    }
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const u32 local = w * WAVE_ITEMS + j * WAVE + lane;
        const bool valid = FULL || local < tile_count;
        const u32 digit = (u32)(key[j] >> shift) & mask;
        // wave64 multisplit: lanes holding the same digit find each other with 8 ballots.  Per bit:
        // s = the bit spread over a word (one v_bfe_i32), the ballot of it, and the lanes that
        // differ from this one in that bit accumulate in `diff` (xor + or per half).  (The kernel is
        // bound by its write requests, then by instruction issue: 9 -> 5 VALU per bit here was worth
        // 10 % of a pass; interleaving the rows' chains to fill the ballot hazards was measured and is
        // slower -- the rows then cannot start before all four loads have landed.)
        u32 diff_lo = 0, diff_hi = 0;
#pragma unroll
        for (int bit = 0; bit < DB; bit++) {
            const u32 sbit = (u32)__builtin_amdgcn_sbfe((int)digit, (u32)bit, 1u);      // 0 or ~0
            const u64 bal = __ballot((int)sbit < 0);
            diff_lo |= (u32)bal ^ sbit;
            diff_hi |= (u32)(bal >> 32) ^ sbit;
        }
        u64 same = ~(((u64)diff_hi << 32) | diff_lo);
        if (!FULL) {
            same &= __ballot(valid);
            if (!valid) same = 1ull << lane;
        }
        const u32 cnt = (u32)__popcll(same);
        // set bits of `same` below this lane (v_mbcnt_lo/hi take a per-lane mask)
        const u32 before = __builtin_amdgcn_mbcnt_hi((u32)(same >> 32), __builtin_amdgcn_mbcnt_lo((u32)same, 0u));
        const int leader = __ffsll((unsigned long long)same) - 1;
        const u32 half = (digit & 1u) * 16u;
        u32 prior = 0;
        if (valid && before == 0) prior = atomicAdd(&lds.wave_cnt[w][digit >> 1], cnt << half);
        prior = (__shfl(prior, leader, WAVE) >> half) & 0xFFFFu;
        slot[j] = prior + before;
    }
    __syncthreads();

    // digit pair e (< 128): wave bases, then the exclusive scan over digits inside the tile
    constexpr int PAIRS = BINS / 2;
    u32 tot_lo = 0, tot_hi = 0;
    if (tid < (u32)PAIRS) {
        u32 run = 0;                                    // both halves at once: the sums stay below 2^16
#pragma unroll
        for (int k = 0; k < WAVES; k++) {
            const u32 c = lds.wave_cnt[k][tid];
            lds.wave_cnt[k][tid] = run;
            run += c;
        }
        tot_lo = run & 0xFFFFu;
        tot_hi = run >> 16;
    }
    const u32 tsum = tot_lo + tot_hi;
    const u32 inc = wave_inclusive_sum(tsum);
    if (lane == 63 && w < (u32)(PAIRS / WAVE)) lds.wsum[w] = inc;
    __syncthreads();
    if (tid < (u32)PAIRS) {
        u32 start = inc - tsum;
#pragma unroll
        for (int k = 0; k < PAIRS / WAVE; k++)
            if ((u32)k < w) start += lds.wsum[k];
        lds.digit_start[2 * tid] = start;
        lds.global_base[2 * tid] = out_base.x - start;
        start += tot_lo;
        lds.digit_start[2 * tid + 1] = start;
        lds.global_base[2 * tid + 1] = out_base.y - start;
    }
    __syncthreads();

    // keys: scatter into tile order, then stream out as contiguous per-digit runs
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const u32 local = w * WAVE_ITEMS + j * WAVE + lane;
        if (FULL || local < tile_count) {
            const u32 digit = (u32)(key[j] >> shift) & mask;
            slot[j] = lds.digit_start[digit] + ((lds.wave_cnt[w][digit >> 1] >> ((digit & 1u) * 16u)) & 0xFFFFu) + slot[j];
            lds.s_keys[slot[j]] = key[j];
        }
    }
    __syncthreads();
    u32 dst[IPT];
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const u32 pos = j * THREADS + tid;
        if (FULL || pos < tile_count) {
            const K k = lds.s_keys[pos];
            dst[j] = lds.global_base[(u32)(k >> shift) & mask] + pos;
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

template <class K, class Src>
__global__ __launch_bounds__(RS_THREADS, (RS_IPT <= 4 ? 8 : 4)) void radix_scatter_kernel(
    Src src, K *__restrict__ keys_out, u32 *__restrict__ vals_out, u32 n, int shift, u32 mask,
    const u32 *__restrict__ hist, const u32 *__restrict__ group_prefix, u32 n_tiles, RsSeg seg)
{
    // XCD-aware tile order: workgroups go round-robin over the 8 XCDs, each with its own L2.  The
    // per-digit runs of NEIGHBOURING tiles are neighbours in the output, so neighbouring tiles are
    // given to the same XCD (XCD x takes the x-th eighth of the tiles): the partial lines at the
    // seams of the runs meet in one L2 instead of reaching memory as masked writes from two.
    const u32 per_xcd = (n_tiles + 7u) / 8u;
    const u32 tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    u32 tile_base, tile_count;
    rs_tile_range(seg, n, tile, tile_base, tile_count);
    if (tile_count == 0) return;                        // (segmented: the tiles that fill up a document's last group)
    // every tile but the last (of a document) is full: no bounds tests in its code path
    __shared__ ScatterLds<K> lds;
    if (tile_count == (u32)RS_TILE)
        radix_scatter_tile<K, Src, true>(lds, src, keys_out, vals_out, shift, mask, hist, group_prefix, tile, tile_base, tile_count);
    else
        radix_scatter_tile<K, Src, false>(lds, src, keys_out, vals_out, shift, mask, hist, group_prefix, tile, tile_base, tile_count);
}

template <class K> struct SortBufs {
    K *keys[2];
    u32 *vals[2];
};

struct NoGen {
    static constexpr int MODE = 0;
};

static inline int radix_pass_count(int bits) { return (bits + RS_DB - 1) / RS_DB; }

// element count and pass bookkeeping for bench.py's byte model
static inline void radix_account(Ctx &ctx, size_t key_bytes, u32 n, bool generated)
{
    if (!ctx.stats) return;
    ctx.stats->radix_passes++;
    if (!generated) {                           // (generator passes are accounted under their own kernel names)
        ctx.stats->radix_elems += n;
        if (key_bytes == 8) { ctx.stats->radix_elems_u64 += n; ctx.stats->radix_passes_u64++; }
        else { ctx.stats->radix_elems_u32 += n; ctx.stats->radix_passes_u32++; }
    }
    if ((i64)(key_bytes + 4) > ctx.stats->radix_elem_bytes) ctx.stats->radix_elem_bytes = key_bytes + 4;
}

// Sorts on key bits [begin_bit, bits).  Input in buffers [0] -- or, with a generator, produced on the
// fly by the first pass, which then writes into buffers [0].  Returns the index (0/1) of the buffers
// that hold the sorted pairs.
// check_from_bit: the histogram passes at or above that bit test every key for "the whole wavefront on one bin" (keys
// that arrive sorted on their high bits: group numbers, document numbers); the passes below take the digits as spread out.
template <class K, class Gen = NoGen>
static int radix_sort_pairs(Ctx &ctx, SortBufs<K> &b, u32 n, int bits, int begin_bit = 0, Gen gen = Gen(), int check_from_bit = 0,
                            RsSeg seg = RsSeg())
{
    constexpr bool HAS_GEN = !std::is_same<Gen, NoGen>::value;
    if (n == 0 || bits <= begin_bit) return 0;
    const u32 n_groups = seg.n_docs ? seg.n_groups : ceil_div_u32(ceil_div_u32(n, RS_TILE), RS_GROUP);
    const u32 n_tiles = seg.n_docs ? n_groups * RS_GROUP : ceil_div_u32(n, RS_TILE);
    const size_t total_words = seg.n_docs ? (size_t)seg.n_docs * seg.shards * RS_BINS : (size_t)RS_TOTAL_SHARDS * RS_BINS;
    const size_t mark = ctx.arena->mark();
    u32 *hist = ctx.arena->alloc<u32>((size_t)RS_BINS * n_tiles);
    u32 *group_sum = ctx.arena->alloc<u32>((size_t)RS_BINS * n_groups);
    u32 *group_prefix = ctx.arena->alloc<u32>((size_t)RS_BINS * n_groups);
    u32 *totals = ctx.arena->alloc<u32>(2 * total_words);
    if (!ctx.dry) HIP_CHECK(hipMemsetAsync(totals, 0, 2 * total_words * sizeof(u32), ctx.stream));
    const bool prof = ctx.prof && ctx.prof->enabled;
    int cur = 0, pass = 0;
    bool first = HAS_GEN;                       // the generator pass reads no buffers and writes [0]
    for (int shift = begin_bit; shift < bits; shift += RS_DB, pass++) {
        const u32 mask = (1u << std::min(RS_DB, bits - shift)) - 1u;
        const PairSrc<K> src{b.keys[cur], b.vals[cur]};
        const int out = first ? 0 : cur ^ 1;
        u32 *tot = totals + (pass & 1) * total_words, *tot_next = totals + ((pass & 1) ^ 1) * total_words;
        // (static strings: the profiler keeps the pointers)
        const char *name_hist = sizeof(K) == 8 ? (first ? "radix_hist_kernel<u64,gen>" : "radix_hist_kernel<u64>")
                                               : (first ? "radix_hist_kernel<u32,gen>" : "radix_hist_kernel<u32>");
        const char *name_scatter = sizeof(K) == 8 ? (first ? "radix_scatter_kernel<u64,gen>" : "radix_scatter_kernel<u64>")
                                                  : (first ? "radix_scatter_kernel<u32,gen>" : "radix_scatter_kernel<u32>");
        if (!ctx.dry) {
            if (prof) ctx.prof->begin(name_hist, ctx.stream);
            const bool check = shift + RS_DB > check_from_bit;
            if constexpr (HAS_GEN) {
                if (first && check)
                    hipLaunchKernelGGL((radix_hist_kernel<K, Gen, true>), dim3(n_groups), dim3(RS_HIST_THREADS), 0, ctx.stream, gen, n,
                                       shift, mask, n_tiles, hist, group_sum, tot, seg);
                else if (first)
                    hipLaunchKernelGGL((radix_hist_kernel<K, Gen, false>), dim3(n_groups), dim3(RS_HIST_THREADS), 0, ctx.stream, gen, n,
                                       shift, mask, n_tiles, hist, group_sum, tot, seg);
            }
            if (!first && check)
                hipLaunchKernelGGL((radix_hist_kernel<K, PairSrc<K>, true>), dim3(n_groups), dim3(RS_HIST_THREADS), 0, ctx.stream, src,
                                   n, shift, mask, n_tiles, hist, group_sum, tot, seg);
            else if (!first)
                hipLaunchKernelGGL((radix_hist_kernel<K, PairSrc<K>, false>), dim3(n_groups), dim3(RS_HIST_THREADS), 0, ctx.stream, src,
                                   n, shift, mask, n_tiles, hist, group_sum, tot, seg);
            HIP_CHECK(hipGetLastError());
            if (prof) ctx.prof->end(ctx.stream);
        }
        if (seg.n_docs) {
            if (seg.n_big < seg.n_docs)
                LAUNCH(ctx, radix_spine_docs_kernel, seg.n_docs, (const u32 *)group_sum, (const u32 *)tot, tot_next, group_prefix, seg);
            if (seg.n_big)
                LAUNCH(ctx, radix_spine_kernel, dim3(RS_BINS / RS_SPINE_COLS, seg.n_big), (const u32 *)group_sum, n_groups,
                       (const u32 *)tot, tot_next, group_prefix, seg);
        } else {
            LAUNCH(ctx, radix_spine_kernel, dim3(RS_BINS / RS_SPINE_COLS, 1u), (const u32 *)group_sum, n_groups, (const u32 *)tot,
                   tot_next, group_prefix, seg);
        }
        if (!ctx.dry) {
            if (prof) ctx.prof->begin(name_scatter, ctx.stream);
            const dim3 grid(8 * ((n_tiles + 7) / 8));
            if constexpr (HAS_GEN) {
                if (first)
                    hipLaunchKernelGGL((radix_scatter_kernel<K, Gen>), grid, dim3(RS_THREADS), 0, ctx.stream, gen, b.keys[out],
                                       b.vals[out], n, shift, mask, (const u32 *)hist, (const u32 *)group_prefix, n_tiles, seg);
            }
            if (!first)
                hipLaunchKernelGGL((radix_scatter_kernel<K, PairSrc<K>>), grid, dim3(RS_THREADS), 0, ctx.stream, src, b.keys[out],
                                   b.vals[out], n, shift, mask, (const u32 *)hist, (const u32 *)group_prefix, n_tiles, seg);
            HIP_CHECK(hipGetLastError());
            if (prof) ctx.prof->end(ctx.stream);
        }
        cur = out;
        radix_account(ctx, sizeof(K), n, first);
        first = false;
    }
    ctx.arena->release(mark);
    return cur;
}
