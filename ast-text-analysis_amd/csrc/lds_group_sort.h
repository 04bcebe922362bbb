// lds_group_sort.h -- a refinement round for tie groups that fit a workgroup's LDS.
//
// A round of window_sort.h orders the members of every tie group of its (compacted) domain by the NEXT
// window of symbols.  As a global sort that is a radix sort by (group number, window): eight passes over
// 64-bit keys, three of them spent on group numbers that are in order before the first pass starts.  The
// groups are contiguous stretches of the domain, and in text made of words most of them are small: here a
// workgroup takes the groups that START in its chunk of LG_CHUNK domain positions -- up to LG_CAP elements
// as long as no group is longer than LG_MAX_GROUP --, builds their keys straight from the text, sorts them
// in LDS by (group inside the tile, window) with the wave64 ballot multisplit of radix_sort.h, and writes
// what the round's write-back would: the members in refined order into their slots, the new naming
// predicate, the next domain's elements, and the LCP entries at the seams.  One read of the domain, no key
// ever reaches HBM.  Groups longer than LG_MAX_GROUP are left to the global sort (cover[] says which
// positions were taken).
#pragma once
#include "common.h"

#ifndef LG_THREADS
#define LG_THREADS 1024
#endif
#ifndef LG_IPT
#define LG_IPT 5
#endif
#define LG_CAP (LG_THREADS * LG_IPT)        // 5120 elements per workgroup
#ifndef LG_CHUNK
#define LG_CHUNK 2048                       // a workgroup owns the groups that start in its chunk
#endif
#define LG_MAX_GROUP (LG_CAP - LG_CHUNK - 64)   // 3008: longer groups take the global path
#define LG_WORDS (LG_CAP / 64)              // 80 words of start bits
#define LG_NONE 0xFFFFFFFFu
#define LG_WAVES (LG_THREADS / WAVE)
static_assert(LG_CHUNK / 64 <= WAVE, "the chunk's start words are examined by one wave");
static_assert(LG_MAX_GROUP <= 63 * 64, "the end of the last group is looked for in 64 words");

struct LgLds {
    u64 keys[LG_CAP];
    u32 vals[LG_CAP];
    u32 wave_cnt[LG_WAVES][128];            // per-wave digit counts (16-bit pairs), then wave bases
    u32 digit_start[256];
    u32 wsum[LG_WAVES];
    u64 start_bits[LG_WORDS + 1];           // bit q: domain position chunk_base + q starts a group; after the sort: the NEW starts, by tile position
    u32 word_prefix[LG_WORDS];              // group starts of the tile before the word
    u32 hdr[2];                             // first position of the tile (relative to the chunk), its elements
};

// Is domain position j left to the global sort?  cover[c] = the stretch of positions workgroup c took
// (LG_NONE: no group starts in its chunk).  A position in front of its chunk's first group start belongs
// to the last group of the nearest earlier chunk that has a start -- at most two chunks back, or the group
// is longer than any workgroup takes.
// Inside chunk c the positions left over are two stretches: [a0, a1) in front of the chunk's first group start --
// what the tile of an earlier chunk did not reach --, and [b0, b1) behind the chunk's own tile.
struct LgUncovered {
    const uint2 *cover;
    u32 m;
    __device__ __forceinline__ void rest(u32 c, u32 &a0, u32 &a1, u32 &b0, u32 &b1) const
    {
        const u32 lo = c * LG_CHUNK, hi = lo + LG_CHUNK < m ? lo + LG_CHUNK : m;
        // (the three entries requested at once: a loop that stops at the first chunk with a start was a round trip per step)
        const uint2 own = cover[c], p1 = cover[c >= 1u ? c - 1u : 0u], p2 = cover[c >= 2u ? c - 2u : 0u];
        u32 reached = lo;                               // how far an earlier tile reaches into this chunk
        if (c >= 1u && p1.x != LG_NONE) reached = p1.y > lo ? p1.y : lo;
        else if (c >= 2u && p2.x != LG_NONE) reached = p2.y > lo ? p2.y : lo;
        if (reached > hi) reached = hi;
        a0 = reached;
        a1 = own.x != LG_NONE ? own.x : hi;             // (own.x >= reached: a tile ends where a group ends)
        if (a1 < a0) a1 = a0;
        b0 = own.x != LG_NONE ? (own.y < hi ? own.y : hi) : hi;
        b1 = hi;
    }
    __device__ __forceinline__ u32 operator()(u32 j) const
    {
        if (j >= m) return 0u;
        u32 a0, a1, b0, b1;
        rest(j / LG_CHUNK, a0, a1, b0, b1);
        return ((j >= a0 && j < a1) || (j >= b0 && j < b1)) ? 1u : 0u;
    }
};

__global__ __launch_bounds__(BLOCK) void lg_rest_count_kernel(const uint2 *__restrict__ cover, u32 m, u32 n_chunks,
                                                              u32 *__restrict__ rest_cnt)
{
    const u32 c = blockIdx.x * BLOCK + threadIdx.x;
    if (c > n_chunks) return;
    u32 cnt = 0;
    if (c < n_chunks) {
        u32 a0, a1, b0, b1;
        LgUncovered{cover, m}.rest(c, a0, a1, b0, b1);
        cnt = (a1 - a0) + (b1 - b0);
    }
    rest_cnt[c] = cnt;
}

// the rest of the domain, compacted: rest_pre[c] = left-over positions in front of chunk c
__global__ __launch_bounds__(BLOCK) void lg_rest_compact_kernel(LgUncovered left, const u32 *__restrict__ rest_pre,
                                                                const u32 *__restrict__ elems,
                                                                const u32 *__restrict__ gstart, u32 m,
                                                                u32 *__restrict__ sub_elem, u32 *__restrict__ sub_gstart,
                                                                u32 *__restrict__ full_idx)
{
    const u32 j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m) return;
    const u32 c = j / LG_CHUNK;
    u32 a0, a1, b0, b1;
    left.rest(c, a0, a1, b0, b1);
    u32 o;
    if (j >= a0 && j < a1) o = rest_pre[c] + (j - a0);
    else if (j >= b0 && j < b1) o = rest_pre[c] + (a1 - a0) + (j - b0);
    else return;
    sub_elem[o] = elems[j];
    sub_gstart[o] = gstart[j];
    full_idx[o] = j;
}

// One stable pass of the in-LDS radix sort: the elements of active waves (wave w owns the LG_IPT * 64
// consecutive positions from w * LG_IPT * 64 on, row j = 64 consecutive ones) move to their places by the
// 8-bit digit at `shift`; on return key[] / val[] hold the new occupants of the thread's positions.
// (waves behind the tile's elements hold nothing and only keep the barriers)
__device__ __forceinline__ void lg_radix_pass(LgLds &lds, u64 (&key)[LG_IPT], u32 (&val)[LG_IPT], int shift, bool active)
{
    const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
    for (u32 i = tid; i < LG_WAVES * 128u; i += LG_THREADS) (&lds.wave_cnt[0][0])[i] = 0;   // (all rows: the sums below run over all of them)
    __syncthreads();
    u32 slot[LG_IPT];
    if (active) {
#pragma unroll
        for (int j = 0; j < LG_IPT; j++) {
            const u32 digit = (u32)(key[j] >> shift) & 0xFFu;
            u32 diff_lo = 0, diff_hi = 0;               // (the multisplit of radix_scatter_tile)
#pragma unroll
            for (int bit = 0; bit < 8; bit++) {
                const u32 sbit = (u32)__builtin_amdgcn_sbfe((int)digit, (u32)bit, 1u);
                const u64 bal = __ballot((int)sbit < 0);
                diff_lo |= (u32)bal ^ sbit;
                diff_hi |= (u32)(bal >> 32) ^ sbit;
            }
            const u64 same = ~(((u64)diff_hi << 32) | diff_lo);
            const u32 cnt = (u32)__popcll(same);
            const u32 before = __builtin_amdgcn_mbcnt_hi((u32)(same >> 32), __builtin_amdgcn_mbcnt_lo((u32)same, 0u));
            const int leader = __ffsll((unsigned long long)same) - 1;
            const u32 half = (digit & 1u) * 16u;
            u32 prior = 0;
            if (before == 0) prior = atomicAdd(&lds.wave_cnt[w][digit >> 1], cnt << half);
            prior = (__shfl(prior, leader, WAVE) >> half) & 0xFFFFu;
            slot[j] = prior + before;
        }
    }
    syncthreads_after_lds_atomics();
    u32 tot_lo = 0, tot_hi = 0;
    if (tid < 128u) {
        u32 run = 0;                                    // both halves at once: the sums stay below 2^16
        u32 c[LG_WAVES];                                // (a fixed trip count: the loads go out back to back -- with the loop
#pragma unroll                                          // bounded by the active waves every step waited for its own load)
        for (int k = 0; k < LG_WAVES; k++) c[k] = lds.wave_cnt[k][tid];
#pragma unroll
        for (int k = 0; k < LG_WAVES; k++) {
            lds.wave_cnt[k][tid] = run;
            run += c[k];
        }
        tot_lo = run & 0xFFFFu;
        tot_hi = run >> 16;
    }
    const u32 tsum = tot_lo + tot_hi;
    const u32 inc = wave_inclusive_sum(tsum);
    if (lane == 63 && w < 2u) lds.wsum[w] = inc;
    __syncthreads();
    if (tid < 128u) {
        const u32 start = inc - tsum + (w == 1u ? lds.wsum[0] : 0u);
        lds.digit_start[2 * tid] = start;
        lds.digit_start[2 * tid + 1] = start + tot_lo;
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int j = 0; j < LG_IPT; j++) {
            const u32 digit = (u32)(key[j] >> shift) & 0xFFu;
            const u32 pos = lds.digit_start[digit] + ((lds.wave_cnt[w][digit >> 1] >> ((digit & 1u) * 16u)) & 0xFFFFu) + slot[j];
            lds.keys[pos] = key[j];
            lds.vals[pos] = val[j];
        }
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int j = 0; j < LG_IPT; j++) {
            const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
            key[j] = lds.keys[local];
            val[j] = lds.vals[local];
        }
    }
}

// The classification of the NEXT domain, folded into the round (symbol windows only): the tile's sorted keys are still
// in LDS when the write-back is done, so the new group bounds are a bit scan away -- a member of a new group of at
// most `limit` suffixes fetches the 8 symbols behind the new depth once, ranks itself against the other members out of
// LDS (window_sort.h: lvl0_place_tied does the same from global arrays) and is placed for good; a member of a larger
// group sets its bit in keep[] (zeroed by the host; OR-ed into, the tile's positions do not start at a word).  What
// dc3_refine_classify_kernel did for these positions in a pass of its own -- reading elements, flags and slots back
// from HBM, a dozen scattered flag reads per tied element -- is gone; that kernel still runs for what the global sort
// took, for prefix-doubling rounds (their names are updated in between) and for the endgame.
// keep == nullptr: off.  A comparison longer than max_len raises *fail (the host restores the domain and repeats
// the classification as mark + commit with the stand-alone kernel).
struct LgClassify {
    u64 *keep = nullptr;
    u32 *fail = nullptr;
    u32 limit = 0, max_len = 0;
    // (not part of the classification, but it travels with it: what the group of a domain position shares beyond `depth`
    // -- first-level keys of variable-length code words, window_sort.h -- or nullptr)
    const uint8_t *xdep = nullptr;
};

// 64 bits of a bit array from bit `pos` on (the array carries one spare word)
__device__ __forceinline__ u64 lg_bits_from(const u64 *bits, u32 pos)
{
    const u32 wi = pos >> 6, sh = pos & 63u;
    const u64 lo = bits[wi] >> sh;
    return sh ? lo | (bits[wi + 1] << (64u - sh)) : lo;
}

// elems / gstart / slots: the round's compacted domain (m elements; gstart[j] != 0: j starts a group).
// Key of an element = (its group's number inside the tile << w2*b) | the w2 symbols at offset `depth` of its
// suffix, as dc3_refine_keys_kernel builds them (zeros behind a terminator; rep_t / ones / highs: the
// terminator test on the window fields).  name_of != nullptr (prefix doubling): the key part is the 32-bit name of
// the suffix `depth` symbols further on instead, as dc3_double_keys_kernel builds it (no terminators, no LCP seams).
__global__ __launch_bounds__(LG_THREADS) void refine_lds_sort_kernel(
    const uint8_t *__restrict__ s8, const u32 *__restrict__ elems, const u32 *__restrict__ gstart,
    const u32 *__restrict__ slots, u32 m, u32 n0, u32 depth, int w2, int b, u32 term_first, u64 rep_t, u64 ones, u64 highs,
    u32 *__restrict__ order_g, u32 *__restrict__ names_g, u32 *__restrict__ elem_out, u32 *__restrict__ flag_out,
    u32 *__restrict__ lcp_g, uint2 *__restrict__ cover, const u32 *__restrict__ name_of, LgClassify cls)
{
    __shared__ LgLds lds;
    const u32 lane = lane_id(), w = wave_id();
    const u32 base = blockIdx.x * LG_CHUNK;
    // ---- the group starts of the window [base, base + LG_CAP) (position m counts as one) ----
    // (a wave's five words: all five loads requested before the first ballot -- one round trip instead of five; a lane
    // behind the domain reads its last entry)
    {
        static_assert(LG_WORDS % LG_WAVES == 0, "every wave takes the same number of words");
        constexpr int PER = LG_WORDS / LG_WAVES;
        u32 gs[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const u64 p = (u64)base + (w + (u32)i * LG_WAVES) * 64u + lane;
            gs[i] = gstart[p < m ? p : (u64)m - 1u];
        }
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const u32 word = w + (u32)i * LG_WAVES;
            const u64 p = (u64)base + word * 64u + lane;
            const bool st = p < m ? gs[i] != 0u : p == m;
            const u64 bal = __ballot(st);
            if (lane == 0) lds.start_bits[word] = bal;
        }
    }
    __syncthreads();
    // ---- wave 0: the tile = from the first start of the chunk to the end of the last group that starts in it ----
    if (w == 0) {
        const u64 wv = lane < LG_CHUNK / 64 ? lds.start_bits[lane] : 0ull;
        const u64 nzb = __ballot(wv != 0ull);
        u32 begin_q = LG_NONE, end_q = LG_NONE;
        if (nzb) {
            const u32 fl = (u32)__ffsll((unsigned long long)nzb) - 1u, ll = 63u - (u32)__builtin_clzll(nzb);
            begin_q = fl * 64u + (u32)__builtin_ctzll(lds.start_bits[fl]);
            const u32 last_q = ll * 64u + 63u - (u32)__builtin_clzll(lds.start_bits[ll]);
            const u32 wi = ll + lane;
            u64 x = wi < LG_WORDS ? lds.start_bits[wi] : 0ull;
            if (lane == 0) x &= ~(((u64)2 << (last_q & 63u)) - 1ull);           // the starts behind the last one of the chunk
            const u64 nb = __ballot(x != 0ull);
            u32 next_q = LG_NONE;
            if (nb) {
                const u32 l2 = (u32)__ffsll((unsigned long long)nb) - 1u;
                const u64 xw = ((u64)__shfl((u32)(x >> 32), l2, WAVE) << 32) | __shfl((u32)x, l2, WAVE);
                next_q = (ll + l2) * 64u + (u32)__builtin_ctzll(xw);
            }
            end_q = (next_q != LG_NONE && next_q - last_q <= LG_MAX_GROUP) ? next_q : last_q;
        }
        if (lane == 0) {
            lds.hdr[0] = begin_q;
            lds.hdr[1] = begin_q == LG_NONE ? 0u : end_q - begin_q;
            cover[blockIdx.x] = begin_q == LG_NONE ? uint2{LG_NONE, LG_NONE} : uint2{base + begin_q, base + end_q};
        }
    }
    __syncthreads();
    const u32 begin_q = lds.hdr[0], n_act = lds.hdr[1];
    if (n_act == 0) return;
    // ---- number the tile's groups: starts in front of each word ----
    if (w == 0) {
        u32 run = 0;
        for (u32 k = 0; k < LG_WORDS; k += WAVE) {
            const u32 word = k + lane;
            u64 x = word < LG_WORDS ? lds.start_bits[word] : 0ull;
            if (word == begin_q >> 6) x &= ~(((u64)1 << (begin_q & 63u)) - 1ull);
            if (word < begin_q >> 6) x = 0ull;
            if (word < LG_WORDS) lds.start_bits[word] = x;                        // (the starts in front of the tile are of no use)
            const u32 c = (u32)__popcll(x);
            const u32 inc = wave_inclusive_sum(c);
            if (word < LG_WORDS) lds.word_prefix[word] = run + inc - c;
            run += __shfl(inc, 63, WAVE);
        }
    }
    __syncthreads();
    // groups of the tile: the starts in [begin_q, begin_q + n_act)
    const u32 last = begin_q + n_act - 1u;
    const u32 n_groups = lds.word_prefix[last >> 6] + (u32)__popcll(lds.start_bits[last >> 6] & (((u64)2 << (last & 63u)) - 1ull));
    const int wbits = name_of ? w2 : w2 * b;            // (prefix doubling: w2 = the bits of a name -- a rank, bit_width(n))
    const int bits = wbits + (n_groups > 1u ? 32 - (int)__builtin_clz(n_groups - 1u) : 0);
    const bool active = w * (LG_IPT * WAVE) < n_act;
    // ---- keys from the text ----
    // Three steps, each over all of the thread's rows, with no branch in between: the elements (coalesced), then every
    // gather at once, then the keys -- row by row a thread waited for three dependent loads per row.  (Rows behind the
    // tile's end read its last element and are overwritten.  Measured: 3 % of the kernel.  The same treatment of the slots
    // in the write-back below made the kernel 50 % SLOWER on natural-language text, 1.36 -> 2.06 ms -- same registers, same
    // occupancy, not explained --, so that loop stays row by row.)
    u64 key[LG_IPT];
    u32 val[LG_IPT];
    if (active) {
        u32 gid[LG_IPT], xd[LG_IPT];
#pragma unroll
        for (int j = 0; j < LG_IPT; j++) {
            const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
            const u32 q = begin_q + (local < n_act ? local : n_act - 1u);
            val[j] = elems[base + q];
            xd[j] = cls.xdep ? (u32)cls.xdep[base + q] : 0u;
            gid[j] = lds.word_prefix[q >> 6] + (u32)__popcll(lds.start_bits[q >> 6] & (((u64)2 << (q & 63u)) - 1ull)) - 1u;
        }
        if (name_of) {
            u32 nm[LG_IPT];
#pragma unroll
            for (int j = 0; j < LG_IPT; j++) nm[j] = name_of[val[j] + depth];
#pragma unroll
            for (int j = 0; j < LG_IPT; j++) key[j] = ((u64)gid[j] << wbits) | (u64)nm[j];
        } else {
            u64 lo8[LG_IPT], hi8[LG_IPT];
#pragma unroll
            for (int j = 0; j < LG_IPT; j++) {
                const u32 p = lvl0_pos(val[j], n0) + depth + xd[j];
                __builtin_memcpy(&lo8[j], s8 + p, 8);
                __builtin_memcpy(&hi8[j], s8 + p + 8, 8);
            }
#pragma unroll
            for (int j = 0; j < LG_IPT; j++) {
                u64 k = gid[j];
                bool ended = false;
                for (int i = 0; i < w2; i++) {
                    const u32 byte = (u32)((i < 8 ? lo8[j] >> (8 * i) : hi8[j] >> (8 * (i - 8))) & 0xFFu);
                    const u32 x = ended ? 0u : byte;
                    ended = ended || x == 0xFFu;
                    k = (k << b) | (u64)(x == 0xFFu ? term_first : x);
                }
                key[j] = k;
            }
        }
#pragma unroll
        for (int j = 0; j < LG_IPT; j++) {
            const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
            if (local >= n_act) { key[j] = ~0ull; val[j] = 0u; }      // (padding: behind everything, in every digit)
        }
    }
    for (int shift = 0; shift < bits; shift += 8) lg_radix_pass(lds, key, val, shift, active);
    // ---- what the round's write-back writes, for the tile ----
    const bool fuse = cls.keep != nullptr && !name_of;  // (uniform over the workgroup)
    if (!active && !fuse) return;
    const u32 nd = depth + (u32)w2;                     // what the members of a new group share
    u32 fbits = 0;                                      // bit j: the thread's element j starts a new group
    u32 slot_of[LG_IPT];
    if (active) {
#pragma unroll
        for (int j = 0; j < LG_IPT; j++) {
            const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
            slot_of[j] = 0;
            if (local >= n_act) { fbits |= 1u << j; continue; }       // (position n_act counts as a start)
            const u64 k = key[j], kp = local ? lds.keys[local - 1u] : 0ull;
            const u64 x = k ^ rep_t;
            const u64 tz = (x - ones) & ~x & highs;
            const u32 r = base + begin_q + local, slot = slots[r], e = val[j];
            const u32 f = (local == 0u || tz != 0ull || k != kp) ? 1u : 0u;
            fbits |= f << j;
            slot_of[j] = slot;
            if (!fuse) order_g[slot] = e;               // (fused: below, once it is known who is placed elsewhere)
            if (names_g) names_g[slot] = f;
            elem_out[r] = e;
            flag_out[r] = f;
            if (lcp_g && f && local > 0u && (k >> wbits) == (kp >> wbits)) {       // a seam inside a group (see dc3_refine_writeback_kernel)
                const u64 d = (k ^ kp) & (((u64)1 << wbits) - 1ull);
                const u32 inv_b = (65536u + (u32)b - 1u) / (u32)b;   // (uniform: x / b = (x * inv_b) >> 16 for bit positions below 64)
                const u32 mism = d ? (u32)w2 - 1u - (((u32)(63 - __builtin_clzll(d)) * inv_b) >> 16) : (u32)w2;
                const u32 term = tz ? (u32)w2 - 1u - (((u32)__builtin_ctzll(tz) * inv_b) >> 16) : (u32)w2;
                lcp_g[slot] = depth + (cls.xdep ? (u32)cls.xdep[r] : 0u) + (mism < term ? mism : term);
            }
        }
    }
    if (!fuse) return;
    // ---- the next domain's classification (LgClassify) ----
#pragma unroll
    for (int j = 0; j < LG_IPT; j++) {                  // the new starts as bits, by tile position (waves behind the tile: all ones)
        const u64 bal = __ballot(!active || ((fbits >> j) & 1u));
        if (lane == 0) lds.start_bits[w * LG_IPT + j] = bal;
    }
    if (threadIdx.x == 0) { lds.start_bits[LG_WORDS] = ~0ull; lds.hdr[0] = 0u; }     // (hdr[0]: the work list's length)
    __syncthreads();                                    // (also: every read of the sorted keys above is done, lds.keys is free)
    // Every element of the tile: untied or left to the next round (its suffix goes to its own slot; a member of a group of
    // more than `limit` sets its keep bit), or a member of a small group: parked in a work list -- the per-wave counters
    // of the sort are idle -- and ranked one member per thread below, so that their text gathers run side by side.
    const u32 limit = cls.limit;
    uint16_t *work = reinterpret_cast<uint16_t *>(&lds.wave_cnt[0][0]);
    constexpr u32 WORK_CAP = sizeof(lds.wave_cnt) / sizeof(uint16_t);
    // the new group [a, bnd) around tile position `local` if it has at most `limit` members, else false
    auto small_group = [&](u32 local, u32 &a, u32 &bnd) -> bool {
        const u32 lo = local >= limit ? local - limit : 0u;
        const u64 left = lg_bits_from(lds.start_bits, lo) & (((u64)2 << (local - lo)) - 1ull);      // positions lo .. local
        const u64 right = lg_bits_from(lds.start_bits, local + 1u) & (((u64)1 << limit) - 1ull);    // local + 1 .. local + limit
        if (!left || !right) return false;
        a = lo + 63u - (u32)__builtin_clzll(left);
        bnd = local + 1u + (u32)__builtin_ctzll(right);
        return bnd - a <= limit;
    };
    u32 pmask = 0;                                      // bit j: the thread's element j is in the work list
#pragma unroll
    for (int j = 0; j < LG_IPT; j++) {
        const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
        bool large = false, parked = false;
        if (active && local < n_act) {
            u32 a = 0, bnd = 0;
            if (!small_group(local, a, bnd)) large = true;
            else parked = bnd - a > 1u;
        }
        // (one LDS atomic per wavefront and row for the list positions)
        const u64 pbal = __ballot(parked);
        u32 at = 0;
        if (pbal) {
            if (lane == (u32)__builtin_ctzll(pbal)) at = atomicAdd(&lds.hdr[0], (u32)__popcll(pbal));
            at = __shfl(at, __builtin_ctzll(pbal), WAVE) + __builtin_amdgcn_mbcnt_hi((u32)(pbal >> 32), __builtin_amdgcn_mbcnt_lo((u32)pbal, 0u));
        }
        if (parked) { pmask |= 1u << j; if (at < WORK_CAP) work[at] = (uint16_t)local; }
        else if (active && local < n_act) order_g[slot_of[j]] = val[j];
        const u64 bal = __ballot(large);                // one keep bit per position of the row
        if (lane == 0 && bal) {
            const u64 g0 = (u64)base + begin_q + w * (LG_IPT * WAVE) + j * WAVE;
            const u32 sh = (u32)(g0 & 63u);
            atomicOr((unsigned long long *)&cls.keep[g0 >> 6], (unsigned long long)(bal << sh));
            if (sh) atomicOr((unsigned long long *)&cls.keep[(g0 >> 6) + 1], (unsigned long long)(bal >> (64u - sh)));
        }
    }
    syncthreads_after_lds_atomics();
    u32 todo = lds.hdr[0];
    if (todo > WORK_CAP) {
        // (more members of small groups than the list holds -- nearly the whole tile: all of them are left to the next
        // round instead, as if their groups were large; a group is ordered here whole or not at all)
#pragma unroll
        for (int j = 0; j < LG_IPT; j++) {
            const bool mine = (pmask >> j) & 1u;
            if (mine) order_g[slot_of[j]] = val[j];
            const u64 bal = __ballot(mine);
            if (lane == 0 && bal) {
                const u64 g0 = (u64)base + begin_q + w * (LG_IPT * WAVE) + j * WAVE;
                const u32 sh = (u32)(g0 & 63u);
                atomicOr((unsigned long long *)&cls.keep[g0 >> 6], (unsigned long long)(bal << sh));
                if (sh) atomicOr((unsigned long long *)&cls.keep[(g0 >> 6) + 1], (unsigned long long)(bal >> (64u - sh)));
            }
        }
        todo = 0;
    }
    for (u32 i = threadIdx.x; i < todo; i += LG_THREADS) {              // the 8 symbols behind the new depth, once per member
        const u32 local = work[i];
        const u32 nde = nd + (cls.xdep ? (u32)cls.xdep[base + begin_q + local] : 0u);
        lds.keys[local] = load_u64_unaligned(s8 + lvl0_pos(lds.vals[local], n0) + nde);
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < todo; i += LG_THREADS) {
        const u32 local = work[i];
        u32 a = 0, bnd = 0;
        (void)small_group(local, a, bnd);
        const u32 e = lds.vals[local], p = lvl0_pos(e, n0);
        const u64 u0 = lds.keys[local];
        const u32 nde = nd + (cls.xdep ? (u32)cls.xdep[base + begin_q + local] : 0u);     // (the members of a group share it)
        u32 r = 0, best = 0;                            // best: longest common prefix with a smaller member
        bool undecided = false;
        for (u32 x = a; x < bnd; x++) {
            if (x == local) continue;
            const u32 p2 = lvl0_pos(lds.vals[x], n0);
            bool decided = false, less = false;         // less: suffix p2 < suffix p
            u32 h = nde;
            for (; h < nde + cls.max_len && !decided; h += 8) {
                const u64 u = h == nde ? u0 : load_u64_unaligned(s8 + p + h);
                const u64 v = h == nde ? lds.keys[x] : load_u64_unaligned(s8 + p2 + h);
                const u64 d = u ^ v, z = ~u;
                const u64 tz = (z - 0x0101010101010101ull) & ~z & 0x8080808080808080ull;
                const u32 mism = d ? (u32)__builtin_ctzll(d) >> 3 : 8u;
                const u32 term = tz ? (u32)__builtin_ctzll(tz) >> 3 : 8u;
                if (term < mism) { less = p2 < p; decided = true; h += term; break; }     // both end in (different) terminators
                if (mism < 8u) { less = ((v >> (8 * mism)) & 0xFFu) < ((u >> (8 * mism)) & 0xFFu); decided = true; h += mism; break; }
            }
            if (!decided) { undecided = true; break; }
            if (less) { r++; best = h > best ? h : best; }
        }
        if (undecided) { atomicOr(cls.fail, 1u); continue; }   // (the host restores the domain and repeats the classification)
        // groups keep their stretches of the global order: the slot of tile position a + r lies a + r - local behind this one's
        const u32 slot = slots[base + begin_q + local];
        const u32 at_g = slot + a + r - local;
        order_g[at_g] = e;
        if (names_g) names_g[slot] = 1;
        if (lcp_g && r > 0) lcp_g[at_g] = best;          // (the first of the group keeps the entry it got when the group split off)
    }
}
