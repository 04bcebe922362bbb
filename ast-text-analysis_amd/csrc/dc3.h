// dc3.h -- suffix array construction on the GPU.
//
// Two constructions produce the one suffix array the reference computes in _kark_sort
// (east/asts/easa.py:155-228); the suffix array is unique for a given symbol order, so the
// result is bit-identical to the reference's suftab although none of its sequential
// dict-counting / list-merging code is reproduced.
//
// A. window_suffix_sort (window_sort.h; text on the byte stream, the path ordinary inputs take):
//      ALL suffixes are keyed by their first w symbols (w*bits <= 32 or 64), generated inside the
//      first pass of one stable LSD radix sort; one classify pass then places every suffix whose
//      key is unique and orders small groups of equal keys directly on the text; members of
//      larger groups are refined in rounds keyed by (group, next window), long repeats by prefix
//      doubling.  Several documents: the document number rides on top of the key.  The LCP table
//      falls out of the sorted keys.  No sample, no ranks, no merge.  In the rare cases it gives up
//      (round limit), for wide alphabets and when memory is short,
//
// B. dc3_suffix_array, data-parallel DC3 / skew (Karkkainen & Sanders 2003), takes over:
//   1. sample positions i mod 3 != 0 become packed keys -- at level 0 on the byte stream the same
//      w-symbol windows and the same classify / refinement as in A, otherwise (s[i],s[i+1],s[i+2])
//      read coalesced from the symbol stream -- sorted by the LDS-staged wave64 radix sort
//      (radix_sort.h)                                           [easa.py:163-167]
//   2. naming = inclusive scan of "key differs from predecessor"  [easa.py:169-182]
//   3. if names are not unique: recurse on the name string, all on device,
//      the host only reads one word per level                  [easa.py:184-190]
//   4. non-sample suffixes: stable compaction of the mod-1 entries of SA12
//      (scan) + one radix sort by first symbol                 [easa.py:192-194]
//   5. merge-path merge of the two sorted sets with the DC3 comparator
//                                                               [easa.py:196-228]
//
// Symbol string convention: s[0..n) in [1, sigma], s[n..n+3) == 0.
#pragma once
#include "common.h"
#include "radix_sort.h"
#include "scan.h"
#include "window_sort.h"
#include <math.h>
#include <algorithm>

// ---- step 1: keys -----------------------------------------------------------
template <class K>
__global__ __launch_bounds__(BLOCK) void dc3_triple_keys_kernel(const u32 *__restrict__ s, u32 n0,
                                                                u32 n02, int b, K *__restrict__ keys,
                                                                u32 *__restrict__ vals)
{
    const u32 t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n02) return;
    const u32 p = dc3_sample_pos(t, n0);
    keys[t] = ((K)s[p] << (2 * b)) | ((K)s[p + 1] << b) | (K)s[p + 2];
    vals[t] = t;
}

// Level 0 of an EASA build: every symbol >= term_first is a unique string
// terminator, ordered by its position in the corpus.  All terminators share the
// code term_first in the key and the symbols behind the first terminator of a
// triple are dropped; triples are enumerated in text order, so the STABLE sort
// leaves equal keys in position order = terminator order.  Triples holding a
// terminator are unique and get their own name (KeyNeqTermIn).  This keeps the
// level-0 keys at 3*bits(sigma_text+1) instead of 3*bits(sigma_text+n_strings).
template <class K>
__global__ __launch_bounds__(BLOCK) void dc3_triple_keys_term_kernel(const u32 *__restrict__ s, u32 n0,
                                                                     u32 n02, int b, u32 term_first,
                                                                     K *__restrict__ keys,
                                                                     u32 *__restrict__ vals)
{
    const u32 u = blockIdx.x * BLOCK + threadIdx.x;
    if (u >= n02) return;
    const u32 q = u >> 1, r = u & 1u;
    const u32 p = 3u * q + 1u + r;                  // sample positions in text order: 1,2,4,5,7,8,...
    u32 c0 = s[p], c1 = s[p + 1], c2 = s[p + 2];
    c0 = c0 < term_first ? c0 : term_first;
    c1 = c0 == term_first ? 0u : (c1 < term_first ? c1 : term_first);
    c2 = (c0 == term_first || c1 == term_first) ? 0u : (c2 < term_first ? c2 : term_first);
    keys[u] = ((K)c0 << (2 * b)) | ((K)c1 << b) | (K)c2;
    vals[u] = r ? n0 + q : q;
}

template <class K> struct KeyNeqTermIn {
    const K *keys;
    int b;
    u32 term_first;
    __device__ __forceinline__ u32 operator()(u32 i) const
    {
        const K k = keys[i];
        const u32 mask = (1u << b) - 1u;
        const bool has_term = ((u32)(k >> (2 * b)) & mask) == term_first ||
                              ((u32)(k >> b) & mask) == term_first || ((u32)k & mask) == term_first;
        return (i == 0 || has_term || k != keys[i - 1]) ? 1u : 0u;
    }
};

// wide alphabets (3b > 64): stage A sorts by the third symbol ...
__global__ __launch_bounds__(BLOCK) void dc3_third_keys_kernel(const u32 *__restrict__ s, u32 n0,
                                                               u32 n02, u32 *__restrict__ keys,
                                                               u32 *__restrict__ vals)
{
    const u32 t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n02) return;
    keys[t] = s[dc3_sample_pos(t, n0) + 2];
    vals[t] = t;
}

// ... stage B re-keys the stage-A order by the packed first two symbols.
__global__ __launch_bounds__(BLOCK) void dc3_pair_keys_kernel(const u32 *__restrict__ s, u32 n0,
                                                              u32 n02, int b,
                                                              const u32 *__restrict__ vals,
                                                              u64 *__restrict__ keys)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n02) return;
    const u32 p = dc3_sample_pos(vals[i], n0);
    keys[i] = ((u64)s[p] << b) | (u64)s[p + 1];
}

__global__ __launch_bounds__(BLOCK) void dc3_gather_third_kernel(const u32 *__restrict__ s, u32 n0,
                                                                 u32 n02,
                                                                 const u32 *__restrict__ vals,
                                                                 u32 *__restrict__ third)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n02) return;
    third[i] = s[dc3_sample_pos(vals[i], n0) + 2];
}

// ---- step 2: naming ----------------------------------------------------------
template <class K> struct KeyNeqIn {          // 1 where a new name starts
    const K *keys;
    __device__ __forceinline__ u32 operator()(u32 i) const
    {
        return (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
    }
};

struct KeyNeq2In {
    const u64 *keys;
    const u32 *third;
    __device__ __forceinline__ u32 operator()(u32 i) const
    {
        return (i == 0 || keys[i] != keys[i - 1] || third[i] != third[i - 1]) ? 1u : 0u;
    }
};

// ---- step 2b: almost-unique names: order the few tied samples directly ---------
// When only a few names are shared, the recursion (a full DC3 over 2n/3 names) is
// replaced by ranking each tied sample inside its group of equal names: the
// suffixes of the name string are compared name by name from offset 1 on (the
// three zero pads end every comparison).  Groups larger than RESOLVE_MAX_GROUP or
// comparisons longer than RESOLVE_MAX_LEN raise `fail` and the caller recurses.
#define RESOLVE_MAX_GROUP 32

__global__ __launch_bounds__(BLOCK) void dc3_resolve_ties_kernel(const u32 *__restrict__ sorted_vals,
                                                                 const u32 *__restrict__ names,
                                                                 const u32 *__restrict__ s12, u32 n02,
                                                                 u32 *__restrict__ sa12, u32 *__restrict__ fail)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n02) return;
    const u32 nm = names[i];
    const u32 t = sorted_vals[i];
    const bool left_same = i > 0 && names[i - 1] == nm;
    const bool right_same = i + 1 < n02 && names[i + 1] == nm;
    if (!left_same && !right_same) { sa12[i] = t; return; }
    u32 a = i, b = i + 1;
    while (a > 0 && names[a - 1] == nm && i - a <= RESOLVE_MAX_GROUP) a--;
    while (b < n02 && names[b] == nm && b - i <= RESOLVE_MAX_GROUP) b++;
    if (b - a > RESOLVE_MAX_GROUP) { atomicOr(fail, 1u); return; }
    u32 r = 0;
    for (u32 x = a; x < b; x++) {
        if (x == i) continue;
        const u32 t2 = sorted_vals[x];
        u32 h = 1, c1, c2;
        do {
            c1 = s12[t + h];
            c2 = s12[t2 + h];
            h++;
        } while (c1 == c2 && h <= RESOLVE_MAX_LEN);
        if (c1 == c2) { atomicOr(fail, 1u); return; }
        if (c2 < c1) r++;
    }
    sa12[a + r] = t;
}

// ---- step 3: ranks of the sample suffixes, dense in text order --------------------
// R12[2q]   = rank of the sample suffix at position 3q+1
// R12[2q+1] = rank of the sample suffix at position 3q+2      (1-based, 0 = past the end)
// Dense (no holes for the mod-0 positions): the random 4-byte rank stores land in
// 2n/3 words instead of being strided over an 8-byte-per-position array, which
// lets L2 / Infinity Cache combine them (measured 1.5x faster at 64 Mi symbols,
// 1.5x at 256 Mi).
__device__ __forceinline__ u32 dc3_r12_index(u32 t, u32 n0) { return t < n0 ? 2u * t : 2u * (t - n0) + 1u; }

__global__ __launch_bounds__(BLOCK) void dc3_rank_kernel(const u32 *__restrict__ sa12, u32 n0, u32 n02,
                                                         u32 *__restrict__ r12)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n02) r12[dc3_r12_index(sa12[i], n0)] = i + 1;
}


// Beyond the Infinity Cache (R12 > ~192 MB) the random 4-byte stores above each cost a
// read-modify-write of a 64-byte sector in HBM.  Then the (slot, rank) pairs are first
// bucketed by the top 8 bits of the slot -- one stable radix pass -- so that the stores of a
// bucket land in a window of R12 that fits the L2.
__global__ __launch_bounds__(BLOCK) void dc3_rank_pairs_kernel(const u32 *__restrict__ sa12, u32 n0, u32 n02,
                                                               u32 *__restrict__ slots, u32 *__restrict__ ranks)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n02) return;
    slots[i] = dc3_r12_index(sa12[i], n0);
    ranks[i] = i + 1;
}

__global__ __launch_bounds__(BLOCK) void dc3_rank_store_kernel(const u32 *__restrict__ slots,
                                                               const u32 *__restrict__ ranks, u32 n02,
                                                               u32 *__restrict__ r12)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n02) r12[slots[i]] = ranks[i];
}

// SR[p] = (s[p], R[p]) with R[p] = 0 for p mod 3 == 0 and past the end: symbols and
// ranks interleaved in text order, so that ONE 24-byte window SR[p..p+2] holds
// everything the merge comparator needs for the suffix at p:
// (s[p], s[p+1], R[p+1], R[p+2]).  Streaming: reads s and R12 in order.
__global__ __launch_bounds__(BLOCK) void dc3_interleave_kernel(const u32 *__restrict__ s,
                                                               const u32 *__restrict__ r12, u32 n_pad,
                                                               uint2 *__restrict__ sr)
{
    const u32 p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= n_pad) return;
    const u32 q = p / 3u, m = p - 3u * q;
    sr[p] = make_uint2(s[p], m == 0 ? 0u : r12[2u * q + m - 1u]);
}

// ---- step 4: non-sample suffixes --------------------------------------------
struct LtIn {                                   // 1 for the mod-1 entries of SA12
    const u32 *sa12;
    u32 n0;
    __device__ __forceinline__ u32 operator()(u32 i) const { return sa12[i] < n0 ? 1u : 0u; }
};

__global__ __launch_bounds__(BLOCK) void dc3_compact_s0_kernel(const u32 *__restrict__ s,
                                                               const u32 *__restrict__ sa12,
                                                               const u32 *__restrict__ slot, u32 n0,
                                                               u32 n02, u32 *__restrict__ keys,
                                                               u32 *__restrict__ vals)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n02) return;
    const u32 t = sa12[i];
    if (t < n0) {
        const u32 p0 = 3u * t;                  // the mod-0 position in front of sample 3t+1
        keys[slot[i]] = s[p0];
        vals[slot[i]] = p0;
    }
}

// Level 0 with the byte stream: the non-sample suffixes are sorted on the one-byte
// class code (0xFF = terminator) in a single radix pass.  Suffixes that START with a
// terminator all carry 0xFF and end up as one block at the end of SA0; their true
// order is terminator order = position order, so that block is simply rewritten
// with the terminator-holding mod-0 positions in increasing order (a compaction).
__global__ __launch_bounds__(BLOCK) void dc3_compact_s0_bytes_kernel(const uint8_t *__restrict__ s8,
                                                                     const u32 *__restrict__ sa12,
                                                                     const u32 *__restrict__ slot, u32 n0,
                                                                     u32 n02, u32 *__restrict__ keys,
                                                                     u32 *__restrict__ vals)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n02) return;
    const u32 t = sa12[i];
    if (t < n0) {
        const u32 p0 = 3u * t;
        keys[slot[i]] = s8[p0];
        vals[slot[i]] = p0;
    }
}

struct TermAtMod0In {                           // 1 iff mod-0 position 3q holds a terminator; defined on [0, n0]
    const uint8_t *s8;
    u32 n0;
    __device__ __forceinline__ u32 operator()(u32 q) const { return (q < n0 && s8[3u * q] == 0xFFu) ? 1u : 0u; }
};

__global__ __launch_bounds__(BLOCK) void dc3_term_block_kernel(const uint8_t *__restrict__ s8,
                                                               const u32 *__restrict__ ex, u32 n0,
                                                               u32 *__restrict__ sa0)
{
    const u32 q = blockIdx.x * BLOCK + threadIdx.x;
    if (q >= n0 || s8[3u * q] != 0xFFu) return;
    sa0[n0 - ex[n0] + ex[q]] = 3u * q;          // ex[n0] = number of terminator-first non-sample suffixes
}

// ---- step 5: merge -----------------------------------------------------------
// Merge-path merge of A = SA12 (sample suffixes, minus the dummy) and B = SA0.
// The DC3 comparator [easa.py:202-207] on self-contained tuples:
//   sample at p, p mod 3 == 1:  (s[p], R[p+1])          vs (s[j], R[j+1])
//   sample at p, p mod 3 == 2:  (s[p], s[p+1], R[p+2])  vs (s[j], s[j+1], R[j+2])
struct MergeTup {
    u32 w0, w1, r1, r2;
};

__device__ __forceinline__ MergeTup dc3_load_tup(const uint2 *__restrict__ sr, u32 p)
{
    const uint2 x0 = sr[p], x1 = sr[p + 1], x2 = sr[p + 2];
    MergeTup t;
    t.w0 = x0.x; t.w1 = x1.x; t.r1 = x1.y; t.r2 = x2.y;
    return t;
}

__device__ __forceinline__ bool dc3_a_leq_b(const MergeTup &a, bool a_mod1, const MergeTup &b)
{
    if (a.w0 != b.w0) return a.w0 < b.w0;
    if (a_mod1) return a.r1 <= b.r1;
    if (a.w1 != b.w1) return a.w1 < b.w1;
    return a.r2 <= b.r2;
}

#define MERGE_IPT 4
#define MERGE_TILE (BLOCK * MERGE_IPT)     // outputs per workgroup

// splits[i] = number of A elements among the first i*MERGE_TILE outputs
__global__ __launch_bounds__(BLOCK) void dc3_merge_partition_kernel(const uint2 *__restrict__ sr,
                                                                    const u32 *__restrict__ A, u32 nA,
                                                                    const u32 *__restrict__ B, u32 nB,
                                                                    u32 n0, u32 n_tiles,
                                                                    u32 *__restrict__ splits)
{
    const u32 tile = blockIdx.x * BLOCK + threadIdx.x;
    if (tile > n_tiles) return;
    const u64 kk = (u64)tile * MERGE_TILE;
    const u32 k = kk < (u64)nA + nB ? (u32)kk : nA + nB;
    u32 lo = k > nB ? k - nB : 0u;
    u32 hi = k < nA ? k : nA;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        const u32 t = A[mid];
        const MergeTup a = dc3_load_tup(sr, dc3_sample_pos(t, n0));
        const MergeTup b = dc3_load_tup(sr, B[k - 1u - mid]);
        if (dc3_a_leq_b(a, t < n0, b)) lo = mid + 1u; else hi = mid;
    }
    splits[tile] = lo;
}

__global__ __launch_bounds__(BLOCK) void dc3_merge_tile_kernel(const uint2 *__restrict__ sr,
                                                               const u32 *__restrict__ A, u32 nA,
                                                               const u32 *__restrict__ B, u32 nB, u32 n0,
                                                               const u32 *__restrict__ splits,
                                                               u32 *__restrict__ sa_out)
{
    __shared__ u32 l_w0[MERGE_TILE], l_w1[MERGE_TILE], l_r1[MERGE_TILE], l_r2[MERGE_TILE];
    __shared__ u32 l_pos[MERGE_TILE];       // text position; bit 31 = "sample with p mod 3 == 1"
    __shared__ u32 l_out[MERGE_TILE];
    const u32 tid = threadIdx.x;
    const u32 n = nA + nB;
    const u32 k0 = blockIdx.x * MERGE_TILE;
    const u32 count = n - k0 < (u32)MERGE_TILE ? n - k0 : (u32)MERGE_TILE;
    const u32 a0 = splits[blockIdx.x], a1 = splits[blockIdx.x + 1];
    const u32 na = a1 - a0, b0 = k0 - a0;
    const u32 nb = count - na;

    // stage the tile: coalesced reads of the two sorted lists, one 24-byte gather per suffix
#pragma unroll
    for (int j = 0; j < MERGE_IPT; j++) {
        const u32 idx = j * BLOCK + tid;
        if (idx < count) {
            u32 p, flag = 0;
            if (idx < na) {
                const u32 t = A[a0 + idx];
                p = dc3_sample_pos(t, n0);
                flag = t < n0 ? 0x80000000u : 0u;
            } else {
                p = B[b0 + (idx - na)];
            }
            const MergeTup tp = dc3_load_tup(sr, p);
            l_w0[idx] = tp.w0; l_w1[idx] = tp.w1; l_r1[idx] = tp.r1; l_r2[idx] = tp.r2;
            l_pos[idx] = p | flag;
        }
    }
    __syncthreads();

    // per-thread merge path inside the tile
    const u32 d0 = tid * MERGE_IPT;
    if (d0 < count) {
        u32 lo = d0 > nb ? d0 - nb : 0u;
        u32 hi = d0 < na ? d0 : na;
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            const u32 y = na + (d0 - 1u - mid);
            const MergeTup a = {l_w0[mid], l_w1[mid], l_r1[mid], l_r2[mid]};
            const MergeTup b = {l_w0[y], l_w1[y], l_r1[y], l_r2[y]};
            if (dc3_a_leq_b(a, (l_pos[mid] >> 31) != 0, b)) lo = mid + 1u; else hi = mid;
        }
        u32 x = lo, y = d0 - lo;               // heads: A[x], B[y]
        MergeTup ta = {0, 0, 0, 0}, tb = {0, 0, 0, 0};
        u32 pa = 0, pb = 0;
        if (x < na) { ta = MergeTup{l_w0[x], l_w1[x], l_r1[x], l_r2[x]}; pa = l_pos[x]; }
        if (y < nb) { const u32 q = na + y; tb = MergeTup{l_w0[q], l_w1[q], l_r1[q], l_r2[q]}; pb = l_pos[q]; }
#pragma unroll
        for (int i = 0; i < MERGE_IPT; i++) {
            if (d0 + i < count) {
                bool take_a;
                if (y >= nb) take_a = true;
                else if (x >= na) take_a = false;
                else take_a = dc3_a_leq_b(ta, (pa >> 31) != 0, tb);
                if (take_a) {
                    l_out[d0 + i] = pa & 0x7FFFFFFFu;
                    x++;
                    if (x < na) { ta = MergeTup{l_w0[x], l_w1[x], l_r1[x], l_r2[x]}; pa = l_pos[x]; }
                } else {
                    l_out[d0 + i] = pb;
                    y++;
                    if (y < nb) { const u32 q = na + y; tb = MergeTup{l_w0[q], l_w1[q], l_r1[q], l_r2[q]}; pb = l_pos[q]; }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MERGE_IPT; j++) {
        const u32 idx = j * BLOCK + tid;
        if (idx < count) sa_out[k0 + idx] = l_out[idx];
    }
}

// ---- step 5b: merge + LCP in one pass (level 0 of a single-document byte-stream build) ----
// REC[p] = { s8[p..p+7] (8 symbols, symbol p in the low byte), R[p+1], R[p+2] }: ONE aligned
// 16-byte gather per suffix feeds both the merge comparator (symbols p, p+1 and the two
// ranks) and the LCP of neighbouring outputs (8-symbol windows staged in LDS), which removes
// the second random pass over the symbol stream that a separate LCP kernel needs.
// Symbols are byte codes: 0xFF = terminator class, two 0xFF are different terminators
// ordered by position (comparator) and end the common prefix (LCP).
__global__ __launch_bounds__(BLOCK) void dc3_records_kernel(const uint8_t *__restrict__ s8,
                                                            const u32 *__restrict__ r12, u32 n,
                                                            uint4 *__restrict__ rec)
{
    const u32 p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= n) return;
    u64 win;
    __builtin_memcpy(&win, s8 + p, 8);
    const u32 q = p / 3u, m = p - 3u * q;
    u32 r1 = 0, r2 = 0;
    if (m == 0) { r1 = r12[2u * q]; r2 = r12[2u * q + 1u]; }
    else if (m == 1) r1 = r12[2u * q + 1u];
    else r2 = r12[2u * q + 2u];
    rec[p] = make_uint4((u32)win, (u32)(win >> 32), r1, r2);
}

// a: sample suffix at pa (a_mod1: pa mod 3 == 1), b: non-sample suffix at pb
__device__ __forceinline__ bool dc3_rec_a_leq_b(u32 a_lo, u32 a_r1, u32 a_r2, bool a_mod1, u32 pa,
                                                u32 b_lo, u32 b_r1, u32 b_r2, u32 pb)
{
    const u32 a0 = a_lo & 0xFFu, b0 = b_lo & 0xFFu;
    if (a0 != b0) return a0 < b0;
    if (a0 == 0xFFu) return pa < pb;
    if (a_mod1) return a_r1 <= b_r1;
    const u32 a1 = (a_lo >> 8) & 0xFFu, b1 = (b_lo >> 8) & 0xFFu;
    if (a1 != b1) return a1 < b1;
    if (a1 == 0xFFu) return pa < pb;
    return a_r2 <= b_r2;
}

__global__ __launch_bounds__(BLOCK) void dc3_merge_partition_rec_kernel(const uint4 *__restrict__ rec,
                                                                        const u32 *__restrict__ A, u32 nA,
                                                                        const u32 *__restrict__ B, u32 nB,
                                                                        u32 n0, u32 n_tiles,
                                                                        u32 *__restrict__ splits)
{
    const u32 tile = blockIdx.x * BLOCK + threadIdx.x;
    if (tile > n_tiles) return;
    const u64 kk = (u64)tile * MERGE_TILE;
    const u32 k = kk < (u64)nA + nB ? (u32)kk : nA + nB;
    u32 lo = k > nB ? k - nB : 0u;
    u32 hi = k < nA ? k : nA;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        const u32 t = A[mid];
        const u32 pa = dc3_sample_pos(t, n0), pb = B[k - 1u - mid];
        const uint4 a = rec[pa], b = rec[pb];
        if (dc3_rec_a_leq_b(a.x, a.z, a.w, t < n0, pa, b.x, b.z, b.w, pb)) lo = mid + 1u; else hi = mid;
    }
    splits[tile] = lo;
}

// common prefix of two suffixes whose first 8 symbols are the windows wa, wb; beyond the
// windows the byte stream is read directly (rare)
__device__ __forceinline__ u32 dc3_window_lcp(u64 wa, u64 wb, const uint8_t *__restrict__ s8, u32 pa, u32 pb,
                                              const LcpBudget &budget)
{
    const u64 d = wa ^ wb, z = ~wa;
    const u64 t = (z - 0x0101010101010101ull) & ~z & 0x8080808080808080ull;
    const u32 mism = d ? (u32)__builtin_ctzll(d) >> 3 : 8u;
    const u32 term = t ? (u32)__builtin_ctzll(t) >> 3 : 8u;
    const u32 step = mism < term ? mism : term;
    if (step < 8u) return step;
    return lcp_bytes_capped(s8, pa, pb, 8u, budget);           // (LCP_PARTIAL_BIT set: finished by lcp_finish_kernel)
}

__global__ __launch_bounds__(BLOCK) void dc3_merge_lcp_tile_kernel(const uint4 *__restrict__ rec,
                                                                   const uint8_t *__restrict__ s8,
                                                                   const u32 *__restrict__ A, u32 nA,
                                                                   const u32 *__restrict__ B, u32 nB, u32 n0,
                                                                   const u32 *__restrict__ splits,
                                                                   u32 *__restrict__ sa_out,
                                                                   u32 *__restrict__ lcp_out,
                                                                   u32 *__restrict__ capped, LcpBudget budget)
{
    __shared__ u32 l_lo[MERGE_TILE], l_hi[MERGE_TILE], l_r1[MERGE_TILE], l_r2[MERGE_TILE];
    __shared__ u32 l_pos[MERGE_TILE];       // text position; bit 31 = "sample with p mod 3 == 1"
    __shared__ u32 l_src[MERGE_TILE];       // staged slot that became output o
    const u32 tid = threadIdx.x;
    const u32 n = nA + nB;
    const u32 k0 = blockIdx.x * MERGE_TILE;
    const u32 count = n - k0 < (u32)MERGE_TILE ? n - k0 : (u32)MERGE_TILE;
    const u32 a0 = splits[blockIdx.x], a1 = splits[blockIdx.x + 1];
    const u32 na = a1 - a0, b0 = k0 - a0;
    const u32 nb = count - na;

#pragma unroll
    for (int j = 0; j < MERGE_IPT; j++) {
        const u32 idx = j * BLOCK + tid;
        if (idx < count) {
            u32 p, flag = 0;
            if (idx < na) {
                const u32 t = A[a0 + idx];
                p = dc3_sample_pos(t, n0);
                flag = t < n0 ? 0x80000000u : 0u;
            } else {
                p = B[b0 + (idx - na)];
            }
            const uint4 r = rec[p];
            l_lo[idx] = r.x; l_hi[idx] = r.y; l_r1[idx] = r.z; l_r2[idx] = r.w;
            l_pos[idx] = p | flag;
        }
    }
    __syncthreads();

    const u32 d0 = tid * MERGE_IPT;
    if (d0 < count) {
        u32 lo = d0 > nb ? d0 - nb : 0u;
        u32 hi = d0 < na ? d0 : na;
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            const u32 y = na + (d0 - 1u - mid);
            const u32 pa = l_pos[mid];
            if (dc3_rec_a_leq_b(l_lo[mid], l_r1[mid], l_r2[mid], (pa >> 31) != 0, pa & 0x7FFFFFFFu,
                                l_lo[y], l_r1[y], l_r2[y], l_pos[y])) lo = mid + 1u; else hi = mid;
        }
        u32 x = lo, y = d0 - lo;               // heads: slot x (A), slot na + y (B)
#pragma unroll
        for (int i = 0; i < MERGE_IPT; i++) {
            if (d0 + i < count) {
                bool take_a;
                if (y >= nb) take_a = true;
                else if (x >= na) take_a = false;
                else {
                    const u32 pa = l_pos[x], q = na + y;
                    take_a = dc3_rec_a_leq_b(l_lo[x], l_r1[x], l_r2[x], (pa >> 31) != 0, pa & 0x7FFFFFFFu,
                                             l_lo[q], l_r1[q], l_r2[q], l_pos[q]);
                }
                if (take_a) { l_src[d0 + i] = x; x++; }
                else { l_src[d0 + i] = na + y; y++; }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MERGE_IPT; j++) {
        const u32 o = j * BLOCK + tid;
        if (o < count) {
            const u32 sb = l_src[o];
            const u32 pb = l_pos[sb] & 0x7FFFFFFFu;
            sa_out[k0 + o] = pb;
            if (lcp_out && o > 0) {         // the first rank of a tile is finished by dc3_lcp_heads_kernel
                const u32 sa_ = l_src[o - 1];
                const u32 pa = l_pos[sa_] & 0x7FFFFFFFu;
                const u32 h = dc3_window_lcp(((u64)l_hi[sa_] << 32) | l_lo[sa_], ((u64)l_hi[sb] << 32) | l_lo[sb],
                                             s8, pa, pb, budget);
                if (h & LCP_PARTIAL_BIT) raise_flag(capped);
                lcp_out[k0 + o] = h;
            }
        }
    }
}

// LCP of the first rank of every merge tile (its left neighbour lives in the previous tile)
__global__ __launch_bounds__(BLOCK) void dc3_lcp_heads_kernel(const uint8_t *__restrict__ s8,
                                                              const u32 *__restrict__ sa, u32 n,
                                                              u32 *__restrict__ lcp, u32 *__restrict__ capped, LcpBudget budget)
{
    const u64 r64 = ((u64)blockIdx.x * BLOCK + threadIdx.x) * MERGE_TILE;
    if (r64 >= n) return;
    const u32 r = (u32)r64;
    if (r == 0) { lcp[0] = 0; return; }
    const u32 pa = sa[r - 1], pb = sa[r];
    u64 wa, wb;
    __builtin_memcpy(&wa, s8 + pa, 8);
    __builtin_memcpy(&wb, s8 + pb, 8);
    const u32 h = dc3_window_lcp(wa, wb, s8, pa, pb, budget);
    if (h & LCP_PARTIAL_BIT) raise_flag(capped);
    lcp[r] = h;
}

// ---- host driver ----------------------------------------------------------------
// s: n+3 symbols (three zero pads), values in [1, sigma].  sa_out: n words.
// term_first > 0 (level 0 of an EASA build only): symbols >= term_first are
// unique terminators in increasing order of position.
// Returns the number of levels executed.
template <class K, class Flag>
static const u32 *dc3_sort_and_name(Ctx &ctx, SortBufs<K> &sb, u32 n02, int bits, u32 *names, Flag make_flag)
{
    const int r = radix_sort_pairs<K>(ctx, sb, n02, bits);
    auto flag = make_flag(sb.keys[r]);
    device_scan<decltype(flag), true>(ctx, flag, n02, names);
    return sb.vals[r];
}

// s8 (level 0 of an EASA build with sigma_text <= 254): the byte stream; `s` is then not read at all.
// lcp_out (with s8 only): also emit the LCP table of the suffix array (single document).
static int dc3_suffix_array(Ctx &ctx, const u32 *s, u32 n, u32 sigma, u32 *sa_out, int depth = 0,
                            u32 term_first = 0, const uint8_t *s8 = nullptr, u32 *lcp_out = nullptr,
                            u32 *lcp_capped = nullptr)
{
    const u32 n0 = (n + 2) / 3, n1 = (n + 1) / 3, n2 = n / 3, n02 = n0 + n2;
    const int b = bit_width_u32(sigma);
    const u32 g02 = ceil_div_u32(n02, BLOCK);
    int levels = 1;
    Arena &ar = *ctx.arena;
    const size_t mark_level = ar.mark();
    u32 *sa12 = ar.alloc<u32>(n02);
    const size_t mark_s12 = ar.mark();
    u32 *s12 = ar.alloc<u32>((size_t)n02 + 3);

    // -- sort the sample triples, name them --------------------------------
    u32 n_names = 0;
    if (ctx.dry && depth == 0 && term_first) {         // sizing run: the byte-stream level 0, priced with 64-bit keys
        const size_t mark = ar.mark();
        u32 unused = 0;
        (void)dc3_level0_bytes<u64>(ctx, nullptr, n0, n02, 5, 12, term_first, sa12, s12, unused);
        ar.release(mark);
    }
    if (s8) {
        const size_t mark = ar.mark();
        const int bt = bit_width_u32(term_first);          // bits of the compressed level-0 alphabet
        const int w = lvl0_window(n, bt, term_first);
        const bool final_order = w * bt <= 32 && !ctx.knobs.force_wide_keys
            ? dc3_level0_bytes<u32>(ctx, s8, n0, n02, w, bt, term_first, sa12, s12, n_names)
            : dc3_level0_bytes<u64>(ctx, s8, n0, n02, w, bt, term_first, sa12, s12, n_names);
        if (final_order) n_names = n02;
        ar.release(mark);
    } else {
        const size_t mark = ar.mark();
        u32 *names = ar.alloc<u32>(n02);
        const u32 *sorted_vals = nullptr;
        const int bt = bit_width_u32(term_first);          // bits of the compressed level-0 alphabet
        if (term_first > 0 && 3 * bt <= 32) {
            SortBufs<u32> sb;
            for (int k = 0; k < 2; k++) { sb.keys[k] = ar.alloc<u32>(n02); sb.vals[k] = ar.alloc<u32>(n02); }
            LAUNCH(ctx, (dc3_triple_keys_term_kernel<u32>), g02, s, n0, n02, bt, term_first, sb.keys[0], sb.vals[0]);
            sorted_vals = dc3_sort_and_name<u32>(ctx, sb, n02, 3 * bt, names, [&](const u32 *k) {
                return KeyNeqTermIn<u32>{k, bt, term_first}; });
        } else if (term_first > 0) {                       // bt <= 21 (at most every code point), so 3*bt <= 63 fits 64 bits
            SortBufs<u64> sb;
            for (int k = 0; k < 2; k++) { sb.keys[k] = ar.alloc<u64>(n02); sb.vals[k] = ar.alloc<u32>(n02); }
            LAUNCH(ctx, (dc3_triple_keys_term_kernel<u64>), g02, s, n0, n02, bt, term_first, sb.keys[0], sb.vals[0]);
            sorted_vals = dc3_sort_and_name<u64>(ctx, sb, n02, 3 * bt, names, [&](const u64 *k) {
                return KeyNeqTermIn<u64>{k, bt, term_first}; });
        } else if (3 * b <= 32) {
            SortBufs<u32> sb;
            for (int k = 0; k < 2; k++) { sb.keys[k] = ar.alloc<u32>(n02); sb.vals[k] = ar.alloc<u32>(n02); }
            LAUNCH(ctx, (dc3_triple_keys_kernel<u32>), g02, s, n0, n02, b, sb.keys[0], sb.vals[0]);
            sorted_vals = dc3_sort_and_name<u32>(ctx, sb, n02, 3 * b, names, [&](const u32 *k) {
                return KeyNeqIn<u32>{k}; });
        } else if (3 * b <= 64) {
            SortBufs<u64> sb;
            for (int k = 0; k < 2; k++) { sb.keys[k] = ar.alloc<u64>(n02); sb.vals[k] = ar.alloc<u32>(n02); }
            LAUNCH(ctx, (dc3_triple_keys_kernel<u64>), g02, s, n0, n02, b, sb.keys[0], sb.vals[0]);
            sorted_vals = dc3_sort_and_name<u64>(ctx, sb, n02, 3 * b, names, [&](const u64 *k) {
                return KeyNeqIn<u64>{k}; });
        } else {
            SortBufs<u32> sa;
            for (int k = 0; k < 2; k++) { sa.keys[k] = ar.alloc<u32>(n02); sa.vals[k] = ar.alloc<u32>(n02); }
            LAUNCH(ctx, dc3_third_keys_kernel, g02, s, n0, n02, sa.keys[0], sa.vals[0]);
            const int ra = radix_sort_pairs<u32>(ctx, sa, n02, b);
            SortBufs<u64> sb;
            sb.keys[0] = ar.alloc<u64>(n02);
            sb.keys[1] = ar.alloc<u64>(n02);
            sb.vals[0] = sa.vals[ra];
            sb.vals[1] = sa.vals[ra ^ 1];
            LAUNCH(ctx, dc3_pair_keys_kernel, g02, s, n0, n02, b, (const u32 *)sb.vals[0], sb.keys[0]);
            const int rb = radix_sort_pairs<u64>(ctx, sb, n02, 2 * b);
            u32 *third = sa.keys[0];
            LAUNCH(ctx, dc3_gather_third_kernel, g02, s, n0, n02, (const u32 *)sb.vals[rb], third);
            device_scan<KeyNeq2In, true>(ctx, KeyNeq2In{sb.keys[rb], third}, n02, names);
            sorted_vals = sb.vals[rb];
        }
        LAUNCH(ctx, dc3_scatter_names_kernel, ceil_div_u32((u64)n02 + 3, BLOCK), sorted_vals,
               (const u32 *)names, n02, s12);
        if (ctx.dry) {
            n_names = n02 > 4 ? n02 - 1 : n02;        // worst case: keep recursing
        } else {
            HIP_CHECK(hipMemcpyAsync(&n_names, names + (n02 - 1), sizeof(u32), hipMemcpyDeviceToHost,
                                     ctx.stream));
            HIP_CHECK(sync_stream(ctx.stream));
            if (n_names == 0 || n_names > n02)
                east_throw(EAST_HIP_ERR_INTERNAL, "dc3: impossible name count");
        }
        if (n_names == n02 && !ctx.dry) {  // unique names: the sorted order is SA12 already
            HIP_CHECK(hipMemcpyAsync(sa12, sorted_vals, (size_t)n02 * sizeof(u32),
                                     hipMemcpyDeviceToDevice, ctx.stream));
        } else if (!ctx.dry && n02 - n_names <= n02 / 8) {
            // few ties: order them in place instead of recursing
            u32 *fail = ar.alloc<u32>(1);
            u32 h_fail = 0;
            HIP_CHECK(hipMemsetAsync(fail, 0, sizeof(u32), ctx.stream));
            LAUNCH(ctx, dc3_resolve_ties_kernel, g02, sorted_vals, (const u32 *)names, (const u32 *)s12, n02, sa12,
                   fail);
            HIP_CHECK(hipMemcpyAsync(&h_fail, fail, sizeof(u32), hipMemcpyDeviceToHost, ctx.stream));
            HIP_CHECK(sync_stream(ctx.stream));
            if (!h_fail) {
                n_names = n02;             // SA12 is final
                if (ctx.stats) ctx.stats->levels_resolved++;
            }
        }
        ar.release(mark);
    }

    // -- recurse on the name string ----------------------------------------
    if (n_names < n02) levels += dc3_suffix_array(ctx, s12, n02, n_names, sa12, depth + 1);
    ar.release(mark_s12);                              // the name string is dead from here on

    // -- ranks in text order, non-sample suffixes, merge --------------------------
    {
        // (unique names: sa12 is the sorted order itself, so rank = index + 1 either way)
        u32 *r12 = ar.alloc<u32>((size_t)2 * n0 + 4);
        if (!ctx.dry) HIP_CHECK(hipMemsetAsync(r12, 0, ((size_t)2 * n0 + 4) * sizeof(u32), ctx.stream));
        if (((size_t)2 * n0 + 4) * sizeof(u32) > ctx.knobs.rank_bucket_bytes || ctx.dry) {
            const size_t mark = ar.mark();
            SortBufs<u32> rp;
            for (int k = 0; k < 2; k++) { rp.keys[k] = ar.alloc<u32>(n02); rp.vals[k] = ar.alloc<u32>(n02); }
            LAUNCH(ctx, dc3_rank_pairs_kernel, g02, (const u32 *)sa12, n0, n02, rp.keys[0], rp.vals[0]);
            const int top = bit_width_u32(2 * n0 + 1);
            const int rr = radix_sort_pairs<u32>(ctx, rp, n02, top, top > 8 ? top - 8 : 0);
            LAUNCH(ctx, dc3_rank_store_kernel, g02, (const u32 *)rp.keys[rr], (const u32 *)rp.vals[rr], n02, r12);
            ar.release(mark);
        } else {
            LAUNCH(ctx, dc3_rank_kernel, g02, (const u32 *)sa12, n0, n02, r12);
        }
        const bool byte_records = s8 != nullptr;       // level 0 on the byte stream: 16-byte records
        uint2 *sr = nullptr;
        uint4 *rec = nullptr;
        if (byte_records || (ctx.dry && term_first)) rec = ar.alloc<uint4>(n);   // (dry run: the larger of the two)
        if (byte_records) {
            LAUNCH(ctx, dc3_records_kernel, ceil_div_u32(n, BLOCK), s8, (const u32 *)r12, n, rec);
        } else {
            sr = ar.alloc<uint2>((size_t)n + 3);
            LAUNCH(ctx, dc3_interleave_kernel, ceil_div_u32((u64)n + 3, BLOCK), s, (const u32 *)r12, n + 3, sr);
        }
        u32 *slot = ar.alloc<u32>(n02);
        device_scan<LtIn, false>(ctx, LtIn{sa12, n0}, n02, slot);
        SortBufs<u32> s0;
        for (int k = 0; k < 2; k++) { s0.keys[k] = ar.alloc<u32>(n0); s0.vals[k] = ar.alloc<u32>(n0); }
        int r0;
        if (ctx.dry && term_first) (void)ar.alloc<u32>((size_t)n0 + 1);   // the byte path's extra scan buffer
        if (s8) {
            LAUNCH(ctx, dc3_compact_s0_bytes_kernel, g02, s8, (const u32 *)sa12, (const u32 *)slot, n0, n02,
                   s0.keys[0], s0.vals[0]);
            r0 = radix_sort_pairs<u32>(ctx, s0, n0, 8);
            u32 *ex = ar.alloc<u32>((size_t)n0 + 1);
            device_scan<TermAtMod0In, false>(ctx, TermAtMod0In{s8, n0}, n0 + 1, ex);
            LAUNCH(ctx, dc3_term_block_kernel, ceil_div_u32(n0, BLOCK), s8, (const u32 *)ex, n0, s0.vals[r0]);
        } else {
            LAUNCH(ctx, dc3_compact_s0_kernel, g02, s, (const u32 *)sa12, (const u32 *)slot, n0, n02,
                   s0.keys[0], s0.vals[0]);
            r0 = radix_sort_pairs<u32>(ctx, s0, n0, b);
        }
        const u32 skip = n0 - n1, nA = n02 - skip;
        const u32 n_tiles = ceil_div_u32(n, MERGE_TILE);
        u32 *splits = ar.alloc<u32>((size_t)n_tiles + 1);
        if (ctx.stats) ctx.stats->merge_elems += n;
        if (byte_records) {
            LAUNCH(ctx, dc3_merge_partition_rec_kernel, ceil_div_u32((u64)n_tiles + 1, BLOCK), (const uint4 *)rec,
                   (const u32 *)sa12 + skip, nA, (const u32 *)s0.vals[r0], n0, n0, n_tiles, splits);
            LAUNCH(ctx, dc3_merge_lcp_tile_kernel, n_tiles, (const uint4 *)rec, s8, (const u32 *)sa12 + skip, nA,
                   (const u32 *)s0.vals[r0], n0, n0, (const u32 *)splits, sa_out, lcp_out, lcp_capped, ctx.lcp_budget);
            if (lcp_out)
                LAUNCH(ctx, dc3_lcp_heads_kernel, ceil_div_u32(n_tiles, BLOCK), s8, (const u32 *)sa_out, n, lcp_out,
                       lcp_capped, ctx.lcp_budget);
        } else {
            LAUNCH(ctx, dc3_merge_partition_kernel, ceil_div_u32((u64)n_tiles + 1, BLOCK), (const uint2 *)sr,
                   (const u32 *)sa12 + skip, nA, (const u32 *)s0.vals[r0], n0, n0, n_tiles, splits);
            LAUNCH(ctx, dc3_merge_tile_kernel, n_tiles, (const uint2 *)sr, (const u32 *)sa12 + skip, nA,
                   (const u32 *)s0.vals[r0], n0, n0, (const u32 *)splits, sa_out);
        }
    }
    ar.release(mark_level);
    return levels;
}
