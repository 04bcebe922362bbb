// dc3.h -- suffix array construction on the GPU.
//
// Two constructions produce the one suffix array the reference computes in _kark_sort
// (east/asts/easa.py:155-228); the suffix array is unique for a given symbol order, so the
// result is bit-identical to the reference's suftab although none of its sequential
// dict-counting / list-merging code is reproduced.
//
// A. window_suffix_sort (text on the byte stream, the path ordinary inputs take):
//      ALL suffixes are keyed by their first w symbols (w*bits <= 32 or 64), generated inside the
//      first pass of one stable LSD radix sort; one classify pass then places every suffix whose
//      key is unique and orders small groups of equal keys directly on the text; members of
//      larger groups are refined in rounds keyed by (group, next window).  With one document the
//      LCP table falls out of the sorted keys.  No sample, no ranks, no merge -- but the work is
//      only bounded for inputs whose repeats are short, so it gives up when ties persist and
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
#include <math.h>
#include <algorithm>

__device__ __forceinline__ u32 dc3_sample_pos(u32 t, u32 n0)
{
    return t < n0 ? 3u * t + 1u : 3u * (t - n0) + 2u;
}

// Level 0 on the byte stream runs either over the DC3 sample (element t = sample number, n0 > 0)
// or over ALL suffixes (n0 == 0: element t = text position), see window_suffix_sort.
__device__ __forceinline__ u32 lvl0_pos(u32 t, u32 n0) { return n0 ? dc3_sample_pos(t, n0) : t; }

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

// Level 0 with the byte stream: name the sample suffixes by their first w symbols
// (w >= 3, w*bits <= 64) instead of 3.  Any order-preserving name over a window that
// covers the triple keeps DC3 correct -- comparing (name(i), name(i+3), ...) still
// walks the two suffixes left to right -- and with sigma^w >> n almost every name is
// unique, so the whole recursion collapses into the in-place tie resolution below.
// Same terminator rules as dc3_triple_keys_term_kernel.
// The key bits left over below the w full symbols (`spare`) take the top bits of symbol w+1:
// still order-preserving, and it thins the ties out further for free.
template <class K> struct WindowSrc {          // the (window key, element) pairs, generated by the first radix pass
    const uint8_t *s8;
    u32 n0;
    int w, b, spare;
    u32 term_first;
    __device__ __forceinline__ K key(u32 u) const
    {
        const u32 p = n0 ? 3u * (u >> 1) + 1u + (u & 1u) : u;     // text order either way (the sort is stable)
        u64 lo8, hi8;                               // w <= 12 symbols: two unaligned 8-byte loads
        __builtin_memcpy(&lo8, s8 + p, 8);
        __builtin_memcpy(&hi8, s8 + p + 8, 8);
        K key = 0;
        bool ended = false;
        for (int i = 0; i < w; i++) {
            const u32 byte = (u32)((i < 8 ? lo8 >> (8 * i) : hi8 >> (8 * (i - 8))) & 0xFFu);
            const u32 x = ended ? 0u : byte;
            ended = ended || x == 0xFFu;
            key = (key << b) | (K)(x == 0xFFu ? term_first : x);
        }
        if (spare > 0) {
            const u32 byte = (u32)((w < 8 ? lo8 >> (8 * w) : hi8 >> (8 * (w - 8))) & 0xFFu);
            const u32 x = ended ? 0u : (byte == 0xFFu ? term_first : byte);
            key = (key << spare) | (K)(x >> (b - spare));
        }
        return key;
    }
    __device__ __forceinline__ u32 val(u32 u) const { return n0 ? ((u & 1u) ? n0 + (u >> 1) : (u >> 1)) : u; }
    // the low 8 bits of key(u) alone -- all the first pass's histogram needs: only the last one or two
    // symbols of the window (and the partial one) are decoded; where the first terminator sits comes from a
    // zero-byte test on the loaded bytes
    __device__ __forceinline__ u32 low_digit(u32 u) const
    {
        const u32 p = n0 ? 3u * (u >> 1) + 1u + (u & 1u) : u;
        u64 lo8, hi8;
        __builtin_memcpy(&lo8, s8 + p, 8);
        __builtin_memcpy(&hi8, s8 + p + 8, 8);
        const u64 zl = ~lo8, zh = ~hi8;                 // zero byte <=> 0xFF
        const u64 tl = (zl - 0x0101010101010101ull) & ~zl & 0x8080808080808080ull;
        const u64 th = (zh - 0x0101010101010101ull) & ~zh & 0x8080808080808080ull;
        const int tpos = tl ? __builtin_ctzll(tl) >> 3 : (th ? 8 + (__builtin_ctzll(th) >> 3) : 16);
        const int m = (8 - spare + b - 1) / b < w ? (8 - spare + b - 1) / b : w;     // full symbols reaching into the low byte
        u32 acc = 0;
        for (int i = w - m; i < w; i++) {
            const u32 byte = (u32)((i < 8 ? lo8 >> (8 * i) : hi8 >> (8 * (i - 8))) & 0xFFu);
            const u32 x = i > tpos ? 0u : byte;
            acc = (acc << b) | (x == 0xFFu ? term_first : x);
        }
        if (spare > 0) {
            const u32 byte = (u32)((w < 8 ? lo8 >> (8 * w) : hi8 >> (8 * (w - 8))) & 0xFFu);
            const u32 x = w > tpos ? 0u : (byte == 0xFFu ? term_first : byte);
            acc = (acc << spare) | (x >> (b - spare));
        }
        return acc & 255u;
    }
};

template <class K> struct KeyNeqWindowIn {
    const K *keys;
    K rep_t, ones, highs;       // terminator code / 1 / top bit replicated into every full-symbol field
    // "does any of the w full symbols equal the terminator code": xor turns such a field into zero,
    // then the zero-field test (x - ones) & ~x & highs (exact for "any field is zero")
    static KeyNeqWindowIn make(const K *keys, int w, int b, int spare, u32 term_first)
    {
        KeyNeqWindowIn f{keys, 0, 0, 0};
        for (int j = 0; j < w; j++) {
            f.rep_t |= (K)term_first << (spare + j * b);
            f.ones |= (K)1 << (spare + j * b);
            f.highs |= (K)1 << (spare + j * b + b - 1);
        }
        return f;
    }
    __device__ __forceinline__ u32 operator()(u32 i) const
    {
        const K k = keys[i];
        const K x = k ^ rep_t;
        const bool has_term = ((x - ones) & ~x & highs) != 0;
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

// s12[t] = name of sample t; also clears the three pad words behind s12.
__global__ __launch_bounds__(BLOCK) void dc3_scatter_names_kernel(const u32 *__restrict__ vals,
                                                                  const u32 *__restrict__ names,
                                                                  u32 n02, u32 *__restrict__ s12)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n02) s12[vals[i]] = names[i];
    if (i < 3) s12[n02 + i] = 0;
}

// ---- step 2b: almost-unique names: order the few tied samples directly ---------
// When only a few names are shared, the recursion (a full DC3 over 2n/3 names) is
// replaced by ranking each tied sample inside its group of equal names: the
// suffixes of the name string are compared name by name from offset 1 on (the
// three zero pads end every comparison).  Groups larger than RESOLVE_MAX_GROUP or
// comparisons longer than RESOLVE_MAX_LEN raise `fail` and the caller recurses.
#define RESOLVE_MAX_GROUP 32
#define RESOLVE_MAX_LEN 2048

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

static size_t g_rank_bucket_bytes = (size_t)192 << 20;      // east_hip_debug_set_rank_bucket_bytes (tests)
static const bool g_trace = getenv("EAST_HIP_TRACE") != nullptr;   // per-round progress on stderr
static bool g_force_wide_keys = false;                      // east_hip_debug_set_window_sort(3) (tests)
static bool g_force_lean = false;                           // east_hip_debug_set_window_sort(2) (tests)
static bool g_window_sort = true;                            // east_hip_debug_set_window_sort (tests)

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
__device__ __forceinline__ u32 dc3_window_lcp(u64 wa, u64 wb, const uint8_t *__restrict__ s8, u32 pa, u32 pb)
{
    u32 h = 0;
    while (true) {
        const u64 d = wa ^ wb, z = ~wa;
        const u64 t = (z - 0x0101010101010101ull) & ~z & 0x8080808080808080ull;
        const u32 mism = d ? (u32)__builtin_ctzll(d) >> 3 : 8u;
        const u32 term = t ? (u32)__builtin_ctzll(t) >> 3 : 8u;
        const u32 step = mism < term ? mism : term;
        h += step;
        if (step < 8u) return h;
        if (h >= LCP_DIRECT_CAP) return LCP_CAP_MARK;          // finished by lcp_finish_kernel
        __builtin_memcpy(&wa, s8 + pa + h, 8);
        __builtin_memcpy(&wb, s8 + pb + h, 8);
    }
}

__global__ __launch_bounds__(BLOCK) void dc3_merge_lcp_tile_kernel(const uint4 *__restrict__ rec,
                                                                   const uint8_t *__restrict__ s8,
                                                                   const u32 *__restrict__ A, u32 nA,
                                                                   const u32 *__restrict__ B, u32 nB, u32 n0,
                                                                   const u32 *__restrict__ splits,
                                                                   u32 *__restrict__ sa_out,
                                                                   u32 *__restrict__ lcp_out,
                                                                   u32 *__restrict__ capped)
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
                                             s8, pa, pb);
                if (h == LCP_CAP_MARK) atomicOr(capped, 1u);
                lcp_out[k0 + o] = h;
            }
        }
    }
}

// LCP of the first rank of every merge tile (its left neighbour lives in the previous tile)
__global__ __launch_bounds__(BLOCK) void dc3_lcp_heads_kernel(const uint8_t *__restrict__ s8,
                                                              const u32 *__restrict__ sa, u32 n,
                                                              u32 *__restrict__ lcp, u32 *__restrict__ capped)
{
    const u64 r64 = ((u64)blockIdx.x * BLOCK + threadIdx.x) * MERGE_TILE;
    if (r64 >= n) return;
    const u32 r = (u32)r64;
    if (r == 0) { lcp[0] = 0; return; }
    const u32 pa = sa[r - 1], pb = sa[r];
    u64 wa, wb;
    __builtin_memcpy(&wa, s8 + pa, 8);
    __builtin_memcpy(&wb, s8 + pb, 8);
    const u32 h = dc3_window_lcp(wa, wb, s8, pa, pb);
    if (h == LCP_CAP_MARK) atomicOr(capped, 1u);
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
// ---- step 2c: refinement of tied names (level 0, byte stream) ---------------------------------
// Natural-language text repeats words and phrases, so half of the suffixes can share their
// w-symbol name.  The tied ones are worked off in rounds, each on the compacted list of what is
// still tied (`domain`: elem[] = the suffixes in their current order, flag[] = 1 where a group of
// equal names starts, slot[] = where each sits in the global order):
//   * small groups (<= REFINE_SMALL_GROUP) are ordered directly on the text, for good;
//   * the members of larger groups are keyed by (group, the NEXT window of symbols) and radix
//     sorted -- the groups stay where they are, their members get ordered by the next symbols --
//     and what is still tied afterwards forms the next, smaller domain.
// A few rounds cover a whole 3-word string.  If ties survive (long repeats: the domain stops
// shrinking), the refined names still are valid DC3 names (order-preserving over a window that covers the triple) and feed
// the recursion; in all-suffix mode the caller falls back to DC3.
#define REFINE_SMALL_GROUP 8
#define REFINE_MAX_ROUNDS 32
#define REFINE_ENDGAME_DOMAIN 262144     // domains this small: groups up to REFINE_ENDGAME_GROUP are ordered directly,
#define REFINE_ENDGAME_GROUP 128         // with comparisons of at most REFINE_ENDGAME_LEN symbols (longer: give up)
#define REFINE_ENDGAME_LEN 256

struct BitIn {                                  // one flag per element, 64 to a word (written by wave ballots)
    const u64 *bits;
    __device__ __forceinline__ u32 operator()(u32 i) const { return (u32)(bits[i >> 6] >> (i & 63u)) & 1u; }
};

struct FlagArrIn {                              // the naming predicate of a compacted domain
    const u32 *flags;
    __device__ __forceinline__ u32 operator()(u32 i) const { return flags[i]; }
};

// LCP of sorted neighbours from their window keys k (the later one) and kp alone: the leading symbol
// fields the two keys have in common, cut at the first terminator field (equal terminator codes
// are two DIFFERENT terminators).  Returns w with whole = true when the full windows agree and hold
// no terminator -- only then does the text have to be read, from offset w on.
template <class K>
__device__ __forceinline__ u32 lvl0_lcp_of_key_pair(const KeyNeqWindowIn<K> &f, int w, int b, int spare, K k, K kp,
                                                    bool &whole)
{
    const u64 d = (u64)(k ^ kp);
    u32 mism = (u32)w;                                   // leading symbol fields in common
    if (d) {
        const int hb = 63 - __builtin_clzll(d);
        if (hb >= spare) mism = (u32)(w - 1 - (hb - spare) / b);
    }
    const K x = k ^ f.rep_t;
    const u64 tz = (u64)((K)(x - f.ones) & ~x & f.highs);    // at most one field holds the terminator code
    const u32 term = tz ? (u32)(w - 1 - (__builtin_ctzll(tz) - spare) / b) : (u32)w;
    const u32 h = mism < term ? mism : term;
    whole = h == (u32)w && d == 0;
    return h;
}

template <class K>
__device__ __forceinline__ u32 lvl0_lcp_of_keys(const KeyNeqWindowIn<K> &f, int w, int b, int spare, u32 r, bool &whole)
{
    return lvl0_lcp_of_key_pair(f, w, b, spare, f.keys[r], f.keys[r - 1], whole);
}

// A tied element j of a domain (elem[], naming predicate `starts`, slot[] or identity): if its group
// of equal names is small (<= REFINE_SMALL_GROUP), its rank inside the group is found by comparing
// the suffixes themselves from offset `depth` on, 8 symbols per step, and it is placed for good:
// order_g, names_g (sample mode) and -- keyed first domain, one document -- its LCP entry, which is
// the longest common prefix with a smaller member found on the way, or comes from the keys for the
// first of the group (lcp_first).  Returns 1 when the group is large: left to the radix round.
template <class Starts, class LcpFirst>
__device__ __forceinline__ u32 lvl0_place_tied(u32 j, const u32 *__restrict__ elem, const Starts &starts,
                                               const u32 *__restrict__ slot, u32 m, const uint8_t *__restrict__ s8,
                                               u32 n0, u32 depth, u32 *__restrict__ order_g,
                                               u32 *__restrict__ names_g, u32 *__restrict__ lcp_g, LcpFirst lcp_first,
                                               u32 *__restrict__ fail, u32 limit = REFINE_SMALL_GROUP,
                                               u32 max_len = RESOLVE_MAX_LEN, u32 *__restrict__ name_of = nullptr)
{
    const bool first = slot == nullptr;
    const u32 e = elem[j];
    u32 a = j, bnd = j + 1;
    while (a > 0 && !starts(a) && j - a <= limit) a--;
    while (bnd < m && !starts(bnd) && bnd - j <= limit) bnd++;
    if (bnd - a > limit) {
        if (first) { order_g[j] = e; if (names_g) names_g[j] = starts(j); }
        return 1;
    }
    const u32 p = lvl0_pos(e, n0);
    u32 r = 0, best = 0;                                // best: longest common prefix with a smaller member
    for (u32 x = a; x < bnd; x++) {
        if (x == j) continue;
        const u32 p2 = lvl0_pos(elem[x], n0);
        bool decided = false, less = false;             // less: suffix p2 < suffix p
        u32 h = depth;
        for (; h < depth + max_len && !decided; h += 8) {
            const u64 u = load_u64_unaligned(s8 + p + h), v = load_u64_unaligned(s8 + p2 + h);
            const u64 d = u ^ v, z = ~u;
            const u64 tz = (z - 0x0101010101010101ull) & ~z & 0x8080808080808080ull;
            const u32 mism = d ? (u32)__builtin_ctzll(d) >> 3 : 8u;
            const u32 term = tz ? (u32)__builtin_ctzll(tz) >> 3 : 8u;
            if (term < mism) { less = p2 < p; decided = true; h += term; break; }     // both end in (different) terminators
            if (mism < 8u) { less = ((v >> (8 * mism)) & 0xFFu) < ((u >> (8 * mism)) & 0xFFu); decided = true; h += mism; break; }
        }
        if (!decided) { atomicOr(fail, 1u); return 0; } // (the host restores the domain and gives up on it)
        if (less) { r++; best = h > best ? h : best; }
    }
    const u32 at = a + r;                               // final place inside the domain
    const u32 at_g = first ? at : slot[at];
    order_g[at_g] = e;
    if (names_g) names_g[first ? j : slot[j]] = 1;
    if (name_of) name_of[p] = at_g;                     // (prefix doubling: a placed suffix is named by its exact position)
    if (lcp_g) lcp_g[at] = r > 0 ? best : lcp_first(at);
    return 0;
}

struct NoLcp {
    __device__ __forceinline__ u32 operator()(u32) const { return 0u; }
};

// First domain = the whole sorted input, 4 consecutive elements per thread (16-byte loads and
// stores).  A suffix whose key differs from both neighbours' (or holds a terminator) is final:
// suffix array entry, name flag (sample mode) and LCP entry (lcp_g: all-suffix mode, one document;
// from the two keys, no text is read) are written at once; tied ones go through lvl0_place_tied.
// keep[]: one bit per element, set for members of large groups; block_keep: their number per workgroup.
#define PLACE_IPT 4
template <class K>
__global__ __launch_bounds__(BLOCK) void lvl0_place_kernel(KeyNeqWindowIn<K> f, const u32 *__restrict__ vals, u32 m,
                                                          const uint8_t *__restrict__ s8, u32 n0, int w, int b,
                                                          int spare, u32 *__restrict__ order_g,
                                                          u32 *__restrict__ names_g, u32 *__restrict__ lcp_g,
                                                          u64 *__restrict__ keep, u32 *__restrict__ block_keep,
                                                          u32 *__restrict__ fail)
{
    __shared__ u32 keep_bits[BLOCK * PLACE_IPT / 32];
    __shared__ u32 work[BLOCK * PLACE_IPT];             // the tied elements of this workgroup's stretch
    __shared__ u32 n_keep, n_work;
    if (threadIdx.x < BLOCK * PLACE_IPT / 32) keep_bits[threadIdx.x] = 0;
    if (threadIdx.x == 0) { n_keep = 0; n_work = 0; }
    __syncthreads();
    const u32 j0 = (blockIdx.x * BLOCK + threadIdx.x) * PLACE_IPT;
    if (j0 < m) {
        // keys j0-1 .. j0+4 (the arrays carry 8 spare entries behind m), elements j0 .. j0+3
        K k[PLACE_IPT + 2];
        k[0] = j0 > 0 ? f.keys[j0 - 1] : (K)0;
#pragma unroll
        for (int e = 0; e < PLACE_IPT; e++) k[e + 1] = f.keys[j0 + e];
        k[PLACE_IPT + 1] = f.keys[j0 + PLACE_IPT];
        u32 v[PLACE_IPT];
#pragma unroll
        for (int e = 0; e < PLACE_IPT; e++) v[e] = vals[j0 + e];
        bool start[PLACE_IPT + 1];
#pragma unroll
        for (int e = 0; e <= PLACE_IPT; e++) {
            const K x = k[e + 1] ^ f.rep_t;
            const bool has_term = ((x - f.ones) & ~x & f.highs) != 0;
            start[e] = j0 + e == 0 || j0 + e >= m || has_term || k[e + 1] != k[e];
        }
        bool all_final = j0 + PLACE_IPT <= m;
#pragma unroll
        for (int e = 0; e < PLACE_IPT; e++) all_final = all_final && start[e] && start[e + 1];
        if (all_final) {
            uint4 o = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<uint4 *>(order_g + j0) = o;
            if (names_g) *reinterpret_cast<uint4 *>(names_g + j0) = uint4{1u, 1u, 1u, 1u};
            if (lcp_g) {
                u32 h[PLACE_IPT];
#pragma unroll
                for (int e = 0; e < PLACE_IPT; e++) {
                    bool whole;
                    h[e] = j0 + e > 0 ? lvl0_lcp_of_key_pair(f, w, b, spare, k[e + 1], k[e], whole) : 0u;
                }
                *reinterpret_cast<uint4 *>(lcp_g + j0) = uint4{h[0], h[1], h[2], h[3]};
            }
        } else {
#pragma unroll
            for (int e = 0; e < PLACE_IPT; e++) {
                const u32 j = j0 + e;
                if (j >= m) break;
                if (start[e] && start[e + 1]) {
                    order_g[j] = v[e];
                    if (names_g) names_g[j] = 1;
                    if (lcp_g) {
                        bool whole;
                        lcp_g[j] = j > 0 ? lvl0_lcp_of_key_pair(f, w, b, spare, k[e + 1], k[e], whole) : 0u;
                    }
                } else if (k[e] == k[e + 1] && k[e + 2] == k[e + 1] &&           // (tied on both sides: worth two more reads)
                           ((j >= REFINE_SMALL_GROUP && f.keys[j - REFINE_SMALL_GROUP] == k[e + 1]) ||
                            (j + REFINE_SMALL_GROUP < m && f.keys[j + REFINE_SMALL_GROUP] == k[e + 1]))) {
                    // sorted keys: an equal key 8 places away means more than 8 equal keys around j -- a large
                    // group (natural-language text: half of the suffixes), flagged without the exact bounds
                    order_g[j] = v[e];
                    if (names_g) names_g[j] = start[e];
                    const u32 local = threadIdx.x * PLACE_IPT + e;
                    atomicOr(&keep_bits[local >> 5], 1u << (local & 31u));
                    atomicAdd(&n_keep, 1u);
                } else {
                    work[atomicAdd(&n_work, 1u)] = j;    // phase 2
                }
            }
        }
    }
    __syncthreads();
    // phase 2: the tied elements, one per thread, so that their text gathers run side by side
    // instead of one after the other inside the thread that met them
    const u32 todo = n_work;
    for (u32 i = threadIdx.x; i < todo; i += BLOCK) {
        const u32 j = work[i];
        auto lcp_first = [&](u32 at) -> u32 {
            bool whole;
            return at > 0 ? lvl0_lcp_of_keys(f, w, b, spare, at, whole) : 0u;
        };
        if (lvl0_place_tied(j, vals, f, (const u32 *)nullptr, m, s8, n0, (u32)w, order_g, names_g, lcp_g, lcp_first, fail)) {
            const u32 local = j - blockIdx.x * (BLOCK * PLACE_IPT);
            atomicOr(&keep_bits[local >> 5], 1u << (local & 31u));
            atomicAdd(&n_keep, 1u);
        }
    }
    __syncthreads();
    // (entries m.. are 0: the exclusive scan over m + 1 entries yields the total)
    if (threadIdx.x < BLOCK * PLACE_IPT / 64)
        keep[(size_t)blockIdx.x * (BLOCK * PLACE_IPT / 64) + threadIdx.x] =
            ((u64)keep_bits[2 * threadIdx.x + 1] << 32) | keep_bits[2 * threadIdx.x];
    if (threadIdx.x == 0) block_keep[blockIdx.x] = n_keep;
}

// A later, compacted domain of m elements (slot[] = where each sits in the global order): untied
// elements already have their place (written by the previous round's write-back); small groups are
// placed by lvl0_place_tied; members of large groups are marked in keep[] (one bit each).
__global__ __launch_bounds__(BLOCK) void dc3_refine_classify_kernel(const u32 *__restrict__ elem, FlagArrIn starts,
                                                                    const u32 *__restrict__ slot, u32 m,
                                                                    const uint8_t *__restrict__ s8, u32 n0, u32 depth,
                                                                    u32 *__restrict__ order_g, u32 *__restrict__ names_g,
                                                                    u64 *__restrict__ keep, u32 *__restrict__ fail,
                                                                    u32 limit, u32 max_len, u32 *__restrict__ name_of)
{
    const u32 j = blockIdx.x * BLOCK + threadIdx.x;
    u32 my_keep = 0;
    if (j < m) {
        const bool left_same = j > 0 && !starts(j);
        const bool right_same = j + 1 < m && !starts(j + 1);
        if (left_same || right_same)
            my_keep = lvl0_place_tied(j, elem, starts, slot, m, s8, n0, depth, order_g, names_g, (u32 *)nullptr, NoLcp(),
                                      fail, limit, max_len, name_of);
    }
    // keep[]: one bit per element (entries m.. of the last word are 0: the exclusive scan over m + 1 yields the total)
    const u64 bal = __ballot(my_keep != 0);
    if (lane_id() == 0) keep[j >> 6] = bal;
}

// A repeat too long for the direct ordering leaves its group half written: put the whole domain back
// the way it was before the classify pass (elem[] and the domain's predicate are untouched by it).
template <class Starts>
__global__ __launch_bounds__(BLOCK) void dc3_refine_restore_kernel(const u32 *__restrict__ elem, Starts starts,
                                                                   const u32 *__restrict__ slot, u32 m,
                                                                   u32 *__restrict__ order_g, u32 *__restrict__ names_g)
{
    const u32 j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m) return;
    const u32 g = slot ? slot[j] : j;
    order_g[g] = elem[j];
    if (names_g) names_g[g] = starts(j);
}

// the members of large groups, compacted: the radix round's input
template <class Starts>
__global__ __launch_bounds__(BLOCK) void dc3_refine_compact_kernel(const u32 *__restrict__ elem, Starts starts,
                                                                   const u32 *__restrict__ slot, BitIn keep,
                                                                   const u32 *__restrict__ idx, u32 m,
                                                                   u32 *__restrict__ slot_out, u32 *__restrict__ elem_out,
                                                                   u32 *__restrict__ group_start)
{
    const u32 j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m || !keep(j)) return;
    const u32 k = idx[j];
    slot_out[k] = slot ? slot[j] : j;
    elem_out[k] = elem[j];
    group_start[k] = starts(j);
}

// key = (dense group number << w2*b) | next window of w2 symbols at offset `depth`
__global__ __launch_bounds__(BLOCK) void dc3_refine_keys_kernel(const uint8_t *__restrict__ s8,
                                                                const u32 *__restrict__ elems,
                                                                const u32 *__restrict__ group, u32 n_tied, u32 n0,
                                                                u32 depth, int w2, int b, u32 term_first,
                                                                u64 *__restrict__ keys, u32 *__restrict__ vals)
{
    const u32 j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n_tied) return;
    const u32 p = lvl0_pos(elems[j], n0) + depth;
    u64 lo8, hi8;
    __builtin_memcpy(&lo8, s8 + p, 8);
    __builtin_memcpy(&hi8, s8 + p + 8, 8);
    u64 key = group[j];
    bool ended = false;
    for (int i = 0; i < w2; i++) {
        const u32 byte = (u32)((i < 8 ? lo8 >> (8 * i) : hi8 >> (8 * (i - 8))) & 0xFFu);
        const u32 x = ended ? 0u : byte;
        ended = ended || x == 0xFFu;
        key = (key << b) | (u64)(x == 0xFFu ? term_first : x);
    }
    keys[j] = key;
    vals[j] = j;
}

// ---- prefix doubling for long repeats (all-suffix mode) ------------------------------------------
// A domain that shrinks slowly holds long repeats (boilerplate, runs of one character): symbol
// windows would take 6-12 symbols off per round.  Then every suffix gets a NAME -- the global
// position where its group of equal prefixes starts, its own position once it is placed -- kept in
// name_of[text position], and a round keys the members of a group by the name of the suffix `depth`
// symbols further on (Manber-Myers / Larsson-Sadakane): names order suffixes by at least `depth`
// symbols, so every round doubles the depth and the rounds are bounded by log2(n).
__global__ __launch_bounds__(BLOCK) void dc3_names_init_kernel(const u32 *__restrict__ order_g, u32 n,
                                                               u32 *__restrict__ name_of)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) name_of[order_g[i]] = i;
}

// start_slot[k] = global position of the first member of the domain's k-th group (inc = inclusive scan of flag)
__global__ __launch_bounds__(BLOCK) void dc3_group_starts_kernel(const u32 *__restrict__ flag, const u32 *__restrict__ inc,
                                                                 const u32 *__restrict__ slot, u32 m,
                                                                 u32 *__restrict__ start_slot)
{
    const u32 r = blockIdx.x * BLOCK + threadIdx.x;
    if (r < m && flag[r]) start_slot[inc[r] - 1u] = slot[r];
}

// (only_kept: members of groups that were just placed for good keep their exact position as name)
__global__ __launch_bounds__(BLOCK) void dc3_names_update_kernel(const u32 *__restrict__ elem, const u32 *__restrict__ inc,
                                                                 const u32 *__restrict__ start_slot, u32 m,
                                                                 const u64 *__restrict__ only_kept,
                                                                 u32 *__restrict__ name_of)
{
    const u32 r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= m) return;
    if (only_kept && !((only_kept[r >> 6] >> (r & 63u)) & 1u)) return;
    name_of[elem[r]] = start_slot[inc[r] - 1u];
}

// key = (dense group number << 32) | name of the suffix `depth` symbols further on
__global__ __launch_bounds__(BLOCK) void dc3_double_keys_kernel(const u32 *__restrict__ name_of,
                                                                const u32 *__restrict__ elems,
                                                                const u32 *__restrict__ group, u32 m, u32 depth,
                                                                u64 *__restrict__ keys, u32 *__restrict__ vals)
{
    const u32 j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m) return;
    keys[j] = ((u64)group[j] << 32) | (u64)name_of[elems[j] + depth];
    vals[j] = j;
}

// after the sort: the slots of the domain receive their members in refined order (globally and as
// the next domain's elem[]), and the naming predicate is updated
__global__ __launch_bounds__(BLOCK) void dc3_refine_writeback_kernel(const u64 *__restrict__ keys,
                                                                     const u32 *__restrict__ vals,
                                                                     const u32 *__restrict__ slots,
                                                                     const u32 *__restrict__ elems, u32 n_tied,
                                                                     u64 rep_t, u64 ones, u64 highs,
                                                                     u32 *__restrict__ order_g, u32 *__restrict__ names_g,
                                                                     u32 *__restrict__ elem_out, u32 *__restrict__ flag_out)
{
    const u32 r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n_tied) return;
    const u64 k = keys[r];
    const u64 x = k ^ rep_t;
    const bool has_term = ((x - ones) & ~x & highs) != 0;       // a terminator inside the window: unique
    const u32 slot = slots[r], e = elems[vals[r]];
    const u32 f = (r == 0 || has_term || k != keys[r - 1]) ? 1u : 0u;
    order_g[slot] = e;
    if (names_g) names_g[slot] = f;
    elem_out[r] = e;
    flag_out[r] = f;
}

// LCP table straight from the sorted window keys (all-suffix mode, one document): neighbours with
// different keys share exactly the leading symbol fields the two keys have in common, cut at the
// first terminator field (equal terminator codes are two DIFFERENT terminators) -- no text is
// touched.  Only neighbours whose whole window agrees (the tied ones, reordered in place since)
// read their suffixes, from offset w on.  Same cap rule as lcp8_kernel.
template <class K>
__global__ __launch_bounds__(BLOCK) void lvl0_lcp_keys_kernel(KeyNeqWindowIn<K> f, int w, int b, int spare,
                                                              const uint8_t *__restrict__ s8,
                                                              const u32 *__restrict__ sa, u32 n,
                                                              u32 *__restrict__ lcp, u32 *__restrict__ capped)
{
    const u32 r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    if (r == 0) { lcp[0] = 0; return; }
    bool whole;
    u32 h = lvl0_lcp_of_keys(f, w, b, spare, r, whole);
    if (whole) {                                         // the whole window agrees: continue on the text
        const u32 i = sa[r - 1], j = sa[r];
        while (true) {
            const u64 xa = load_u64_unaligned(s8 + i + h), xb = load_u64_unaligned(s8 + j + h);
            const u64 dd = xa ^ xb, z = ~xa;
            const u64 t = (z - 0x0101010101010101ull) & ~z & 0x8080808080808080ull;
            const u32 mm = dd ? (u32)__builtin_ctzll(dd) >> 3 : 8u;
            const u32 tt = t ? (u32)__builtin_ctzll(t) >> 3 : 8u;
            const u32 step = mm < tt ? mm : tt;
            h += step;
            if (step < 8u || h >= LCP_DIRECT_CAP) break;
        }
        if (h >= LCP_DIRECT_CAP) { h = LCP_CAP_MARK; atomicOr(capped, 1u); }
    }
    lcp[r] = h;
}

// Level 0 on the byte stream: the window keys are sorted, then one classify pass places everything
// that is untied or tied in a small group (ordered directly on the text); large groups go through
// refinement rounds (step 2c).  Returns true when sa12 is final (no name string is ever written);
// otherwise the refined names are scanned and scattered into s12 for the recursion (sample mode),
// or the caller falls back to DC3 (all-suffix mode, s12 == nullptr).
template <class K>
static bool dc3_level0_bytes(Ctx &ctx, const uint8_t *s8, u32 n0, u32 n02, int w, int bt, u32 term_first, u32 *sa12,
                             u32 *s12, u32 &n_names, u32 *lcp_out = nullptr, u32 *lcp_capped = nullptr)
{
    Arena &ar = *ctx.arena;
    const u32 g02 = ceil_div_u32((u64)n02 + 1, BLOCK);
    // whole passes are paid for anyway: fill the last digit with the top bits of the next symbol
    const int total = ((w * bt + 7) / 8) * 8;
    const int spare = w < 12 ? std::min(total - w * bt, bt - 1) : 0;
    SortBufs<K> sb;
    // (one spare element each: the idle half serves as scratch after the sort)
    // (8 spare entries: the placement pass reads whole 16-byte groups; >= 64 so that the idle half can hold its scratch)
    const size_t n_alloc = (size_t)(n02 > 56 ? n02 : 56) + 8;
    for (int k = 0; k < 2; k++) { sb.keys[k] = ar.alloc<K>(n_alloc); sb.vals[k] = ar.alloc<u32>(n_alloc); }
    const int r = radix_sort_pairs<K, WindowSrc<K>>(ctx, sb, n02, w * bt + spare, 0,
                                                    WindowSrc<K>{s8, n0, w, bt, spare, term_first});
    const KeyNeqWindowIn<K> starts = KeyNeqWindowIn<K>::make(sb.keys[r], w, bt, spare, term_first);
    const u32 *sorted_vals = sb.vals[r];
    u64 *keep = (u64 *)sb.keys[r ^ 1];                   // n02 + 1 bits, in the keys idle since the sort
    u32 *idx = sb.vals[r ^ 1];                          // n02 + 1 entries, likewise
    u32 *names_g = s12 ? ar.alloc<u32>(n02) : nullptr;  // sample mode: the naming predicate as refined so far
    u32 *fail = ar.alloc<u32>(1);
    const u32 gp = ceil_div_u32((u64)n02 + 1, BLOCK * PLACE_IPT);      // workgroups of the placement pass
    u32 *block_keep = ar.alloc<u32>(gp);
    const u32 nb = ceil_div_u32(gp, SCAN_TILE);
    u32 *block_sums = ar.alloc<u32>(nb);
    if (!ctx.dry) HIP_CHECK(hipMemsetAsync(fail, 0, sizeof(u32), ctx.stream));

    // ---- the whole sorted input as the first domain --------------------------------------
    LAUNCH_NAMED(ctx, "lvl0_place_kernel", (lvl0_place_kernel<K>), gp, starts, sorted_vals, n02, s8, n0, w, bt, spare, sa12,
                 names_g, lcp_out, keep, block_keep, fail);
    LAUNCH(ctx, (scan_reduce_kernel<ArrIn>), nb, ArrIn{block_keep}, gp, block_sums);
    u32 m = n02, m_next = n02, h_fail = 0;              // (sizing run: as if everything were tied)
    if (!ctx.dry) {
        std::vector<u32> h_sums(nb);
        HIP_CHECK(hipMemcpyAsync(h_sums.data(), block_sums, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx.stream));
        HIP_CHECK(hipMemcpyAsync(&h_fail, fail, 4, hipMemcpyDeviceToHost, ctx.stream));
        HIP_CHECK(hipStreamSynchronize(ctx.stream));
        m_next = 0;
        for (u32 x : h_sums) m_next += x;
        if (g_trace)
            fprintf(stderr, "[east_hip] level-0 (%s, %u elements, w = %d): %u in large groups%s\n", n0 ? "sample" : "all suffixes",
                    n02, w, m_next, h_fail ? ", a repeat too long to order directly" : "");
        if (h_fail) LAUNCH(ctx, (dc3_refine_restore_kernel<KeyNeqWindowIn<K>>), g02, sorted_vals, starts, (const u32 *)nullptr, n02, sa12, names_g);
        if (!h_fail && m_next == 0) {                   // everything is in place (and the LCP table written)
            if (ctx.stats) ctx.stats->levels_resolved++;
            return true;
        }
    }

    // ---- refinement rounds on the members of large groups ---------------------------------
    // (ctx.lean: the device is short of memory for the rounds' buffers -- straight on to the recursion / DC3)
    bool done = false;
    if (!ctx.lean && !h_fail) {
        const size_t mark_rounds = ar.mark();
        const u32 cap = n02 + 1;                        // natural-language text: most suffixes can be in large groups
        u32 *ebuf[3], *sbuf[2], *fbuf[2];
        for (auto &e : ebuf) e = ar.alloc<u32>(cap);
        for (auto &e : sbuf) e = ar.alloc<u32>(cap);
        for (auto &e : fbuf) e = ar.alloc<u32>(cap);
        u32 *gstart = ar.alloc<u32>(cap), *group = ar.alloc<u32>(cap);
        SortBufs<u64> rb;
        for (int k = 0; k < 2; k++) { rb.keys[k] = ar.alloc<u64>(cap); rb.vals[k] = ar.alloc<u32>(cap); }
        if (ctx.dry) {                                  // sizing run: the transient buffers of one round
            device_scan<BitIn, false>(ctx, BitIn{keep}, n02 + 1, idx);
            (void)radix_sort_pairs<u64>(ctx, rb, cap, 8);
        }
        u32 *name_of = n0 == 0 ? ar.alloc<u32>(n02) : nullptr;      // all-suffix mode: names for prefix doubling
        bool doubling = false;
        u32 depth = (u32)w;
        const u32 *elem = sorted_vals, *slot = nullptr, *flag = nullptr;
        int e_dom = -1, s_dom = 0, f_dom = 0;           // which of the rotating buffers hold the domain
        bool have_idx = false;                          // idx = exclusive scan of keep over the domain
        int stalled = 0;                                // rounds in a row that placed (almost) nothing
        for (int round = 0; !ctx.dry; round++) {
            if (m_next == 0) { done = true; break; }
            // Strings of a few words dissolve within their own length, a round takes 6-12 symbols off.  A domain
            // that stops shrinking is a long repeat (every round would cost the same again): give up on it.
            stalled = (round > 0 && m_next > m - m / 32) ? stalled + 1 : 0;
            if (round == REFINE_MAX_ROUNDS || (stalled == 2 && !doubling && !name_of)) break;
            const u32 gm = ceil_div_u32((u64)m + 1, BLOCK);
            if (name_of && !doubling && round > 0 && m_next > m / 2) {
                // slow shrinking = long repeats: from here on the depth doubles every round (see above)
                doubling = true;
                LAUNCH(ctx, dc3_names_init_kernel, ceil_div_u32(n02, BLOCK), (const u32 *)sa12, n02, name_of);
                device_scan<ArrIn, true>(ctx, ArrIn{flag}, m, group);
                LAUNCH(ctx, dc3_group_starts_kernel, gm, flag, (const u32 *)group, slot, m, gstart);
                LAUNCH(ctx, dc3_names_update_kernel, gm, elem, (const u32 *)group, (const u32 *)gstart, m, (const u64 *)keep,
                       name_of);
                if (g_trace) fprintf(stderr, "[east_hip]   switching to prefix doubling at depth %u\n", depth);
            }
            if (!have_idx) device_scan<BitIn, false>(ctx, BitIn{keep}, m + 1, idx);
            // compact the members of large groups, number their groups, sort by (group, next window)
            const int e_c = (e_dom + 4) % 3, e_out = (e_dom + 5) % 3;      // the two buffers the domain is not in
            u32 *slot_c = sbuf[s_dom ^ 1];
            if (!slot)
                LAUNCH_NAMED(ctx, "dc3_refine_compact_kernel", (dc3_refine_compact_kernel<KeyNeqWindowIn<K>>), gm, elem,
                             starts, slot, BitIn{keep}, (const u32 *)idx, m, slot_c, ebuf[e_c], gstart);
            else
                LAUNCH_NAMED(ctx, "dc3_refine_compact_kernel", (dc3_refine_compact_kernel<FlagArrIn>), gm, elem,
                             FlagArrIn{flag}, slot, BitIn{keep}, (const u32 *)idx, m, slot_c, ebuf[e_c], gstart);
            m = m_next;
            device_scan<ArrIn, true>(ctx, ArrIn{gstart}, m, group);
            // (every group here has more than REFINE_SMALL_GROUP members: a bound on their number saves a read-back)
            const int gbits = bit_width_u32(m / (REFINE_SMALL_GROUP + 1) + 1);
            const u32 gt = ceil_div_u32((u64)m + 1, BLOCK);
            if (doubling) {
                LAUNCH(ctx, dc3_double_keys_kernel, gt, (const u32 *)name_of, (const u32 *)ebuf[e_c], (const u32 *)group, m,
                       depth, rb.keys[0], rb.vals[0]);
                const int rr = radix_sort_pairs<u64>(ctx, rb, m, 32 + gbits);
                LAUNCH(ctx, dc3_refine_writeback_kernel, gt, (const u64 *)rb.keys[rr], (const u32 *)rb.vals[rr],
                       (const u32 *)slot_c, (const u32 *)ebuf[e_c], m, (u64)0, (u64)0, (u64)0, sa12, names_g, ebuf[e_out],
                       fbuf[f_dom ^ 1]);
                // the members' new names: where their (possibly split) group now starts
                device_scan<ArrIn, true>(ctx, ArrIn{fbuf[f_dom ^ 1]}, m, group);
                LAUNCH(ctx, dc3_group_starts_kernel, gt, (const u32 *)fbuf[f_dom ^ 1], (const u32 *)group, (const u32 *)slot_c, m,
                       gstart);
                LAUNCH(ctx, dc3_names_update_kernel, gt, (const u32 *)ebuf[e_out], (const u32 *)group, (const u32 *)gstart, m,
                       (const u64 *)nullptr, name_of);
                depth *= 2;
            } else {
                const int w2 = std::min(12, (64 - gbits) / bt);
                if (w2 < 1) break;
                LAUNCH(ctx, dc3_refine_keys_kernel, gt, s8, (const u32 *)ebuf[e_c], (const u32 *)group, m, n0, depth, w2, bt,
                       term_first, rb.keys[0], rb.vals[0]);
                const int rr = radix_sort_pairs<u64>(ctx, rb, m, gbits + w2 * bt);
                const KeyNeqWindowIn<u64> f = KeyNeqWindowIn<u64>::make(nullptr, w2, bt, 0, term_first);
                LAUNCH(ctx, dc3_refine_writeback_kernel, gt, (const u64 *)rb.keys[rr], (const u32 *)rb.vals[rr],
                       (const u32 *)slot_c, (const u32 *)ebuf[e_c], m, f.rep_t, f.ones, f.highs, sa12, names_g, ebuf[e_out],
                       fbuf[f_dom ^ 1]);
                depth += (u32)w2;
            }
            e_dom = e_out; s_dom ^= 1; f_dom ^= 1;
            elem = ebuf[e_dom]; slot = sbuf[s_dom]; flag = fbuf[f_dom];
            if (ctx.stats) ctx.stats->refine_rounds++;
            // the new, smaller domain: place what is untied or in small groups now, count the rest
            // a small domain is finished by direct ordering of much larger groups (a round costs ~60 launches
            // however few suffixes are left); the comparisons are kept short so that the work stays bounded
            const bool endgame = m <= REFINE_ENDGAME_DOMAIN;
            LAUNCH(ctx, dc3_refine_classify_kernel, gt, elem, FlagArrIn{flag}, slot, m, s8, n0, depth, sa12, names_g, keep,
                   fail, endgame ? (u32)REFINE_ENDGAME_GROUP : (u32)REFINE_SMALL_GROUP,
                   endgame ? (u32)REFINE_ENDGAME_LEN : (u32)RESOLVE_MAX_LEN, doubling ? name_of : (u32 *)nullptr);
            device_scan<BitIn, false>(ctx, BitIn{keep}, m + 1, idx);
            have_idx = true;
            HIP_CHECK(hipMemcpyAsync(&m_next, idx + m, 4, hipMemcpyDeviceToHost, ctx.stream));
            HIP_CHECK(hipMemcpyAsync(&h_fail, fail, 4, hipMemcpyDeviceToHost, ctx.stream));
            HIP_CHECK(hipStreamSynchronize(ctx.stream));
            if (g_trace)
                fprintf(stderr, "[east_hip]   round %d: domain %u, %u still in large groups, depth %u%s\n", round, m, m_next,
                        depth, h_fail ? ", a repeat too long to order directly" : "");
            if (h_fail) {
                LAUNCH(ctx, (dc3_refine_restore_kernel<FlagArrIn>), gt, elem, FlagArrIn{flag}, slot, m, sa12, names_g);
                break;
            }
        }
        ar.release(mark_rounds);
    }
    if (done) {
        if (ctx.stats) ctx.stats->levels_resolved++;
        if (lcp_out)                                    // (the entries written by the first pass are overwritten)
            LAUNCH_NAMED(ctx, "lvl0_lcp_keys_kernel", (lvl0_lcp_keys_kernel<K>), ceil_div_u32(n02, BLOCK), starts, w, bt,
                         spare, s8, (const u32 *)sa12, n02, lcp_out, lcp_capped);
        return true;
    }
    if (!s12) return false;                            // all-suffix mode: the caller falls back to DC3
    // recursion ahead: names by an inclusive scan of the refined predicate, scattered into the name string
    u32 *names = ar.alloc<u32>(n02);
    device_scan<FlagArrIn, true>(ctx, FlagArrIn{names_g}, n02, names);
    if (ctx.dry) {
        n_names = n02 > 4 ? n02 - 1 : n02;            // worst case: recurse
    } else {
        HIP_CHECK(hipMemcpyAsync(&n_names, names + (n02 - 1), 4, hipMemcpyDeviceToHost, ctx.stream));
        HIP_CHECK(hipStreamSynchronize(ctx.stream));
        if (n_names == 0 || n_names > n02) east_throw(EAST_HIP_ERR_INTERNAL, "dc3: impossible name count");
    }
    LAUNCH(ctx, dc3_scatter_names_kernel, ceil_div_u32((u64)n02 + 3, BLOCK), (const u32 *)sa12, (const u32 *)names, n02,
           s12);
    return false;
}

// Width of the level-0 name window.  Widest window: names almost unique (sigma^w >= 64 n) within
// 64-bit keys; but if the window that still fits 32-bit keys leaves only a few per cent of ties
// (sigma^w >= 4 n), the cheaper sort wins and the tie resolution absorbs the difference.
static int lvl0_window(u32 n, int bt, u32 term_first)
{
    int w = 3;
    const int w_max = 64 / bt < 12 ? 64 / bt : 12;
    double reach = (double)term_first * term_first * term_first;
    while (w < w_max && reach < 64.0 * (double)n) { reach *= term_first; w++; }
    const int w32 = 32 / bt;
    if (w32 >= 3 && w32 < w) {
        // the spare bits of the last digit hold the top bits of one more symbol: that many more buckets
        const int spare32 = std::min(((w32 * bt + 7) / 8) * 8 - w32 * bt, bt - 1);
        const double buckets = spare32 > 0 ? (double)((term_first >> (bt - spare32)) + 1u) : 1.0;
        if (pow((double)term_first, w32) * buckets >= 4.0 * (double)n) w = w32;
    }
    return w;
}

// The fast path for text: ALL n suffixes keyed by their first w symbols, one stable sort, the tied
// ones refined by further windows / ordered directly -- no sample, no ranks, no merge.  Ordinary text
// (few and short repeats) ends here; returns false when the ties do not dissolve (long or many
// repeats), and the caller runs DC3, whose work is bounded whatever the input.
static bool window_suffix_sort(Ctx &ctx, const uint8_t *s8, u32 n, u32 term_first, u32 *sa_out, u32 *lcp_out,
                               u32 *lcp_capped)
{
    Arena &ar = *ctx.arena;
    const size_t mark = ar.mark();
    const int bt = bit_width_u32(term_first);
    const int w = lvl0_window(n, bt, term_first);
    u32 n_names = 0;
    const bool ok = w * bt <= 32 && !g_force_wide_keys
                        ? dc3_level0_bytes<u32>(ctx, s8, 0, n, w, bt, term_first, sa_out, nullptr, n_names, lcp_out, lcp_capped)
                     : dc3_level0_bytes<u64>(ctx, s8, 0, n, w, bt, term_first, sa_out, nullptr, n_names, lcp_out, lcp_capped);
    ar.release(mark);
    return ok;
}

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
        const bool final_order = w * bt <= 32 && !g_force_wide_keys
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
        } else if (term_first > 0) {                       // bt <= 12 always, so 3*bt <= 36 fits 64 bits
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
            HIP_CHECK(hipStreamSynchronize(ctx.stream));
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
            HIP_CHECK(hipStreamSynchronize(ctx.stream));
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
        if (((size_t)2 * n0 + 4) * sizeof(u32) > g_rank_bucket_bytes || ctx.dry) {
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
                   (const u32 *)s0.vals[r0], n0, n0, (const u32 *)splits, sa_out, lcp_out, lcp_capped);
            if (lcp_out)
                LAUNCH(ctx, dc3_lcp_heads_kernel, ceil_div_u32(n_tiles, BLOCK), s8, (const u32 *)sa_out, n, lcp_out,
                       lcp_capped);
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
