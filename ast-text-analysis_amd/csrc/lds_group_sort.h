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

#define LG_THREADS 1024
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
    u64 start_bits[LG_WORDS];               // bit q: domain position chunk_base + q starts a group
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
        const uint2 own = cover[c];
        u32 reached = lo;                               // how far an earlier tile reaches into this chunk
        for (u32 k = 1; k <= 2 && k <= c; k++) {
            const uint2 prev = cover[c - k];
            if (prev.x != LG_NONE) { reached = prev.y > lo ? prev.y : lo; break; }
        }
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
// (n_waves: the active waves, 0 .. n_waves - 1 -- the others hold nothing and only keep the barriers)
__device__ __forceinline__ void lg_radix_pass(LgLds &lds, u64 (&key)[LG_IPT], u32 (&val)[LG_IPT], int shift, bool active,
                                              u32 n_waves)
{
    const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
    for (u32 i = tid; i < n_waves * 128u; i += LG_THREADS) (&lds.wave_cnt[0][0])[i] = 0;
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
        for (u32 k = 0; k < n_waves; k++) {
            const u32 c = lds.wave_cnt[k][tid];
            lds.wave_cnt[k][tid] = run;
            run += c;
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

// elems / gstart / slots: the round's compacted domain (m elements; gstart[j] != 0: j starts a group).
// Key of an element = (its group's number inside the tile << w2*b) | the w2 symbols at offset `depth` of its
// suffix, as dc3_refine_keys_kernel builds them (zeros behind a terminator; rep_t / ones / highs: the
// terminator test on the window fields).  name_of != nullptr (prefix doubling): the key part is the 32-bit name of
// the suffix `depth` symbols further on instead, as dc3_double_keys_kernel builds it (no terminators, no LCP seams).
__global__ __launch_bounds__(LG_THREADS) void refine_lds_sort_kernel(
    const uint8_t *__restrict__ s8, const u32 *__restrict__ elems, const u32 *__restrict__ gstart,
    const u32 *__restrict__ slots, u32 m, u32 n0, u32 depth, int w2, int b, u32 term_first, u64 rep_t, u64 ones, u64 highs,
    u32 *__restrict__ order_g, u32 *__restrict__ names_g, u32 *__restrict__ elem_out, u32 *__restrict__ flag_out,
    u32 *__restrict__ lcp_g, uint2 *__restrict__ cover, const u32 *__restrict__ name_of)
{
    __shared__ LgLds lds;
    const u32 lane = lane_id(), w = wave_id();
    const u32 base = blockIdx.x * LG_CHUNK;
    // ---- the group starts of the window [base, base + LG_CAP) (position m counts as one) ----
    for (u32 word = w; word < LG_WORDS; word += LG_WAVES) {
        const u64 p = (u64)base + word * 64u + lane;
        const bool st = p < m ? gstart[p] != 0u : p == m;
        const u64 bal = __ballot(st);
        if (lane == 0) lds.start_bits[word] = bal;
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
    const int wbits = name_of ? 32 : w2 * b;
    const int bits = wbits + (n_groups > 1u ? 32 - (int)__builtin_clz(n_groups - 1u) : 0);
    const bool active = w * (LG_IPT * WAVE) < n_act;
    // ---- keys from the text ----
    u64 key[LG_IPT];
    u32 val[LG_IPT];
    if (active) {
#pragma unroll
        for (int j = 0; j < LG_IPT; j++) {
            const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
            key[j] = ~0ull;                             // (padding: behind everything, in every digit)
            val[j] = 0u;
            if (local < n_act) {
                const u32 q = begin_q + local;
                const u32 e = elems[base + q];
                const u32 gid = lds.word_prefix[q >> 6] + (u32)__popcll(lds.start_bits[q >> 6] & (((u64)2 << (q & 63u)) - 1ull)) - 1u;
                u64 k = gid;
                if (name_of) {
                    k = (k << 32) | (u64)name_of[e + depth];
                } else {
                    const u32 p = lvl0_pos(e, n0) + depth;
                    u64 lo8, hi8;
                    __builtin_memcpy(&lo8, s8 + p, 8);
                    __builtin_memcpy(&hi8, s8 + p + 8, 8);
                    bool ended = false;
                    for (int i = 0; i < w2; i++) {
                        const u32 byte = (u32)((i < 8 ? lo8 >> (8 * i) : hi8 >> (8 * (i - 8))) & 0xFFu);
                        const u32 x = ended ? 0u : byte;
                        ended = ended || x == 0xFFu;
                        k = (k << b) | (u64)(x == 0xFFu ? term_first : x);
                    }
                }
                key[j] = k;
                val[j] = e;
            }
        }
    }
    const u32 n_waves = (n_act + LG_IPT * WAVE - 1u) / (LG_IPT * WAVE);
    for (int shift = 0; shift < bits; shift += 8) lg_radix_pass(lds, key, val, shift, active, n_waves);
    // ---- what the round's write-back writes, for the tile ----
    if (!active) return;
#pragma unroll
    for (int j = 0; j < LG_IPT; j++) {
        const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
        if (local >= n_act) continue;
        const u64 k = key[j], kp = local ? lds.keys[local - 1u] : 0ull;
        const u64 x = k ^ rep_t;
        const u64 tz = (x - ones) & ~x & highs;
        const u32 r = base + begin_q + local, slot = slots[r], e = val[j];
        const u32 f = (local == 0u || tz != 0ull || k != kp) ? 1u : 0u;
        order_g[slot] = e;
        if (names_g) names_g[slot] = f;
        elem_out[r] = e;
        flag_out[r] = f;
        if (lcp_g && f && local > 0u && (k >> wbits) == (kp >> wbits)) {       // a seam inside a group (see dc3_refine_writeback_kernel)
            const u64 d = (k ^ kp) & (((u64)1 << wbits) - 1ull);
            const u32 mism = d ? (u32)(w2 - 1 - (63 - __builtin_clzll(d)) / b) : (u32)w2;
            const u32 term = tz ? (u32)(w2 - 1 - __builtin_ctzll(tz) / b) : (u32)w2;
            lcp_g[slot] = depth + (mism < term ? mism : term);
        }
    }
}
