// score.h -- keyphrase x document score table (easa.py:26-36, 91-139).
//
// One thread per (keyphrase-suffix, document): the thread walks its suffix
// down the document's annotated suffix array.  The reference follows
// childtab sibling chains (O(#children), up to m terminator leaves under the
// root) and looks annotations up with linear scans; here a node is an SA
// interval [lo, hi] at string depth d, the child for symbol c is the
// sub-interval whose suffixes carry c at offset d (binary search -- symbols
// at a fixed depth are sorted inside an interval), and the annotation of an
// interval is its width (root: n_d - m_d).  Whenever the interval shrinks a
// child node has been entered: nodes += 1, acc += width_child / width_parent.
// Only `symbols` and `suftab` are touched.  Arithmetic is fp64 in the
// reference's association order so results are bit-identical:
//     suffix = ((acc + matched) - nodes) [/ matched];  score = (sum in suffix order) / |q|
// A second kernel reduces the per-suffix results in suffix order.
#pragma once
#include "common.h"

#define Q_NOMATCH 0xFFFFFFFFu
#define KGRAM_MAX_K 3                    // tables built from the finished arrays (mark / search kernels)
#define KGRAM_MAX_BINS 65536u
#ifndef KGRAM_KEYS_MAX_K
#define KGRAM_KEYS_MAX_K 4               // tables marked off the sorted window keys (window_sort.h): one level more
#endif
//               // tables marked off the sorted window keys (window_sort.h): one level more
#define KGRAM_KEYS_MAX_BINS 1048576u

// ---- k-gram bucket tables ---------------------------------------------------------
// Most walks end within the first few symbols, where the
// SA intervals are widest and a binary search costs the most probes.  For small
// text alphabets every document therefore gets a table over all k-grams
// (k <= 3, A = sigma_text + 2 codes: 0 = pad, 1..sigma_text, A-1 = "terminator"):
//   kg[d][g] = first rank of document d whose suffix has a k-gram >= g,  kg[d][A^k] = n_d
// so the interval of a text prefix c_0..c_j (j < k) is
//   [ kg[code * A^(k-1-j)], kg[(code+1) * A^(k-1-j)] )   with code = sum c_i A^(j-i).
// The table is read off the finished arrays: rank r starts a bucket iff lcp[r] < k.
// k-gram code of the suffix at p; symbols behind the first terminator are dropped,
// so all suffixes that agree up to (and including) a terminator class share one code
__device__ __forceinline__ u32 kgram_code(const uint8_t *__restrict__ s8, u32 p, int k, u32 A)
{
    u32 g = 0;
    bool ended = false;
    for (int i = 0; i < k; i++) {
        const u32 b = ended ? 0u : (u32)s8[p + i];
        const u32 c = b == 0xFFu ? A - 1u : (b < A ? b : A - 1u);
        ended = ended || b == 0xFFu;
        g = g * A + c;
    }
    return g;
}

__global__ __launch_bounds__(BLOCK) void kgram_mark_kernel(const u32 *__restrict__ lcp, const u32 *__restrict__ sa,
                                                           const uint8_t *__restrict__ s8,
                                                           const u32 *__restrict__ doc_off, u32 n_docs, u32 n,
                                                           int k, u32 A, u32 bins, u32 *__restrict__ kg)
{
    const u32 lo = blockIdx.y;                    // grid: (ranks of the longest document / BLOCK, documents)
    const u32 r = doc_off[lo] + blockIdx.x * BLOCK + threadIdx.x;
    if (r >= doc_off[lo + 1]) return;
    const u32 L = lcp[r];
    if (L >= (u32)k) return;                      // same k-gram as the rank before
    const u32 p = sa[r];
    // two suffixes that part at two different terminators still share their (class) k-gram:
    // only the first of them opens the bucket
    if (r != doc_off[lo] && s8[p + L] == 0xFFu && s8[sa[r - 1] + L] == 0xFFu) return;
    kg[(size_t)lo * (bins + 1) + kgram_code(s8, p, k, A)] = r - doc_off[lo];
}

// The same for many documents (measured: 256 x 1 MiB 1.51 -> 0.77 ms; one 64 MiB document is faster
// with the kernel above, 0.18 against 0.27 ms -- its text gathers leave the L2 and want every
// thread they can get).
__global__ __launch_bounds__(BLOCK) void kgram_mark_tiled_kernel(const u32 *__restrict__ lcp, const u32 *__restrict__ sa,
                                                                 const uint8_t *__restrict__ s8,
                                                                 const u32 *__restrict__ doc_off, u32 n_docs, u32 n,
                                                                 int k, u32 A, u32 bins, u32 *__restrict__ kg)
{
    // grid: (aligned groups of 4 ranks of the longest document / BLOCK, documents).  Phase 1 streams
    // the LCP values with 16-byte loads and parks the ranks that may open a bucket (lcp < k: a
    // sixth of them on 3-word strings, every suffix that meets a terminator within k symbols) in
    // an LDS work list; phase 2 does their gathers one rank per thread.
    __shared__ u32 work[BLOCK * 4];
    __shared__ u32 n_work;
    if (threadIdx.x == 0) n_work = 0;
    __syncthreads();
    const u32 lo = blockIdx.y;
    const u32 first = doc_off[lo], last = doc_off[lo + 1];
    const u32 r0 = (first & ~3u) + (blockIdx.x * BLOCK + threadIdx.x) * 4u;
    if (r0 < last) {
        const uint4 l4 = *reinterpret_cast<const uint4 *>(lcp + r0);
        const u32 ls[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const u32 r = r0 + e;
            if (r >= first && r < last && ls[e] < (u32)k) work[atomicAdd(&n_work, 1u)] = r;   // (lcp >= k: same k-gram as the rank before)
        }
    }
    __syncthreads();
    const u32 todo = n_work;
    for (u32 i = threadIdx.x; i < todo; i += BLOCK) {
        const u32 r = work[i], L = lcp[r];
        const u32 p = sa[r];
        // two suffixes that part at two different terminators still share their (class) k-gram:
        // only the first of them opens the bucket
        if (r != first && s8[p + L] == 0xFFu && s8[sa[r - 1] + L] == 0xFFu) continue;
        kg[(size_t)lo * (bins + 1) + kgram_code(s8, p, k, A)] = r - first;
    }
}

// Long documents (hundreds of ranks per bucket or more): the table entries are found directly --
// kgram_code is non-decreasing along a document's suffix array (terminators sort above every text
// symbol and all fall into the top class), so kg[d][g] is a lower bound found by binary search,
// one thread per entry: log2(n_d) probes each instead of a pass over all n_d LCP values plus the
// suffix-minimum fill (64 MiB document: 0.22 -> 0.05 ms).
__global__ __launch_bounds__(BLOCK) void kgram_search_kernel(const u32 *__restrict__ sa, const uint8_t *__restrict__ s8,
                                                             const u32 *__restrict__ doc_off, int k, u32 A, u32 bins,
                                                             u32 *__restrict__ kg)
{
    const u32 g = blockIdx.x * BLOCK + threadIdx.x, d = blockIdx.y;
    if (g > bins) return;
    const u32 first = doc_off[d], nd = doc_off[d + 1] - first;
    u32 lo = 0, hi = nd;                          // first rank whose k-gram code is >= g
    if (g == bins) lo = nd;
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (kgram_code(s8, sa[first + mid], k, A) < g) lo = mid + 1; else hi = mid;
    }
    kg[(size_t)d * (bins + 1) + g] = lo;
}

// one workgroup per document: kg[d][g] = min over g' >= g (suffix minimum), kg[d][bins] = n_d
// Every wavefront takes a contiguous quarter of the row, 64 entries a step, coalesced: first the quarter's minimum (the
// loads of a step batch go out together), then -- from the back -- the suffix minima: six shuffles per step and the carry of
// the steps behind it.  (Until round 5 every THREAD took a contiguous stretch, twice: 64 cache lines per load instruction,
// one load in flight per wave, 0.065 ms for the 256 rows of 21 952 entries of configs[2] -- 6 % of its score call.)
__global__ __launch_bounds__(BLOCK) void kgram_fill_kernel(const u32 *__restrict__ doc_off, u32 bins,
                                                           u32 *__restrict__ kg)
{
    __shared__ u32 part[WAVES_PER_BLOCK];
    constexpr u32 BATCH = 8;
    const u32 d = blockIdx.x, lane = lane_id(), w = wave_id();
    u32 *row = kg + (size_t)d * (bins + 1);
    const u32 nd = doc_off[d + 1] - doc_off[d];
    const u32 steps = (bins + WAVES_PER_BLOCK * WAVE - 1) / (WAVES_PER_BLOCK * WAVE);     // 64-entry steps per wave
    const u32 b = w * steps * WAVE, e = b + steps * WAVE < bins ? b + steps * WAVE : bins;  // the wave's entries [b, e)
    u32 m = 0xFFFFFFFFu;
    for (u32 s0 = 0; s0 < steps; s0 += BATCH) {
        u32 x[BATCH];
#pragma unroll
        for (u32 k = 0; k < BATCH; k++) {
            const u32 g = b + (s0 + k) * WAVE + lane;
            x[k] = row[g < e ? g : (bins ? bins - 1u : 0u)];
        }
#pragma unroll
        for (u32 k = 0; k < BATCH; k++) {
            const u32 g = b + (s0 + k) * WAVE + lane;
            if (s0 + k < steps && g < e) m = x[k] < m ? x[k] : m;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { const u32 y = __shfl_xor(m, o, WAVE); m = y < m ? y : m; }
    if (lane == 0) part[w] = m;
    __syncthreads();
    u32 carry = nd;                               // the minimum of everything behind the wave's quarter
    for (u32 k = w + 1; k < WAVES_PER_BLOCK; k++) carry = part[k] < carry ? part[k] : carry;
    for (u32 s1 = steps; s1 > 0; s1 -= (s1 < BATCH ? s1 : BATCH)) {
        const u32 first = s1 < BATCH ? 0u : s1 - BATCH;       // the steps [first, s1), from the back
        u32 x[BATCH];
#pragma unroll
        for (u32 k = 0; k < BATCH; k++) {
            const u32 g = b + (first + k) * WAVE + lane;
            x[k] = row[g < e ? g : (bins ? bins - 1u : 0u)];
        }
#pragma unroll
        for (int k = (int)BATCH - 1; k >= 0; k--) {
            if (first + (u32)k >= s1) continue;
            const u32 g = b + (first + (u32)k) * WAVE + lane;
            u32 v = g < e ? x[k] : 0xFFFFFFFFu;   // inclusive suffix minimum over the lanes >= this one, then the carry
            for (int o = 1; o < WAVE; o <<= 1) {
                const u32 y = __shfl_down(v, o, WAVE);
                if (lane + o < WAVE) v = y < v ? y : v;
            }
            v = v < carry ? v : carry;
            if (g < e) row[g] = v;
            carry = __shfl(v, 0, WAVE);
        }
    }
    if (threadIdx.x == 0) row[bins] = nd;
}

// The same for long rows (tables marked off the window keys have up to a million entries per document), all
// accesses coalesced: minima of 1024-entry chunks, their suffix minima per document, then a reverse
// min-scan inside every chunk seeded with the minimum of everything behind it.
#define KGF_CHUNK 1024
__global__ __launch_bounds__(BLOCK) void kgram_chunk_min_kernel(const u32 *__restrict__ kg, u32 bins,
                                                                u32 n_chunks, u32 *__restrict__ cmin)
{
    __shared__ u32 wmin[WAVES_PER_BLOCK];
    const u32 c = blockIdx.x, d = blockIdx.y;
    const u32 *row = kg + (size_t)d * (bins + 1);
    u32 m = 0xFFFFFFFFu;
#pragma unroll
    for (int e = 0; e < KGF_CHUNK / BLOCK; e++) {
        const u32 g = c * KGF_CHUNK + e * BLOCK + threadIdx.x;
        if (g < bins) { const u32 x = row[g]; m = x < m ? x : m; }
    }
    for (int o = 32; o > 0; o >>= 1) { const u32 y = __shfl_xor(m, o, WAVE); m = y < m ? y : m; }
    if (lane_id() == 0) wmin[wave_id()] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < WAVES_PER_BLOCK; k++) m = wmin[k] < m ? wmin[k] : m;
        cmin[(size_t)d * n_chunks + c] = m;
    }
}

// one workgroup per document: csuf[c] = min over the chunks behind c (n_d behind the last one); 256 chunks
// at a time from the back, a reverse min-scan by wave shuffles with the carry of the tiles behind
__global__ __launch_bounds__(BLOCK) void kgram_chunk_suffix_kernel(const u32 *__restrict__ cmin,
                                                                   const u32 *__restrict__ doc_off, u32 n_chunks,
                                                                   u32 *__restrict__ csuf)
{
    __shared__ u32 wmin[WAVES_PER_BLOCK];
    const u32 d = blockIdx.x;
    const u32 *in = cmin + (size_t)d * n_chunks;
    u32 carry = doc_off[d + 1] - doc_off[d];
    for (u32 hi = n_chunks; hi > 0; hi = hi > BLOCK ? hi - BLOCK : 0) {
        const u32 lo = hi > BLOCK ? hi - BLOCK : 0;
        const u32 c = lo + threadIdx.x;
        const u32 x = c < hi ? in[c] : 0xFFFFFFFFu;
        u32 m = x;                                          // inclusive suffix min over the lanes >= this one
        for (int o = 1; o < WAVE; o <<= 1) {
            const u32 y = __shfl_down(m, o, WAVE);
            if (lane_id() + o < WAVE) m = y < m ? y : m;
        }
        if (lane_id() == 0) wmin[wave_id()] = m;
        __syncthreads();
        u32 behind = carry;
        for (u32 k = wave_id() + 1; k < WAVES_PER_BLOCK; k++) behind = wmin[k] < behind ? wmin[k] : behind;
        const u32 next = __shfl_down(m, 1, WAVE);
        const u32 after = lane_id() + 1 < WAVE ? (next < behind ? next : behind) : behind;
        if (c < hi) csuf[(size_t)d * n_chunks + c] = after;         // exclusive: the chunks behind c
        u32 all = carry;
        for (u32 k = 0; k < WAVES_PER_BLOCK; k++) all = wmin[k] < all ? wmin[k] : all;
        carry = all;
        __syncthreads();
    }
}

__global__ __launch_bounds__(BLOCK) void kgram_chunk_fill_kernel(const u32 *__restrict__ csuf,
                                                                 const u32 *__restrict__ doc_off, u32 bins,
                                                                 u32 n_chunks, u32 *__restrict__ kg)
{
    __shared__ u32 wmin[WAVES_PER_BLOCK];
    const u32 c = blockIdx.x, d = blockIdx.y;
    u32 *row = kg + (size_t)d * (bins + 1);
    const u32 g0 = c * KGF_CHUNK + threadIdx.x * 4u;       // four consecutive entries per thread
    u32 x[4];
#pragma unroll
    for (int e = 0; e < 4; e++) x[e] = g0 + e < bins ? row[g0 + e] : 0xFFFFFFFFu;
    // suffix minima inside the thread, then across the lanes behind it, then across the waves behind it
    x[2] = x[3] < x[2] ? x[3] : x[2];
    x[1] = x[2] < x[1] ? x[2] : x[1];
    x[0] = x[1] < x[0] ? x[1] : x[0];
    u32 m = x[0];                                           // inclusive suffix min over the lanes >= this one
    for (int o = 1; o < WAVE; o <<= 1) {
        const u32 y = __shfl_down(m, o, WAVE);
        if (lane_id() + o < WAVE) m = y < m ? y : m;
    }
    if (lane_id() == 0) wmin[wave_id()] = m;
    __syncthreads();
    u32 behind = csuf[(size_t)d * n_chunks + c];            // everything behind this chunk
    for (u32 k = wave_id() + 1; k < WAVES_PER_BLOCK; k++) behind = wmin[k] < behind ? wmin[k] : behind;
    const u32 next = __shfl_down(m, 1, WAVE);               // lanes behind this one (exclusive)
    const u32 after = lane_id() + 1 < WAVE ? (next < behind ? next : behind) : behind;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const u32 v = x[e] < after ? x[e] : after;
        if (g0 + e < bins) row[g0 + e] = v;
    }
    if (c == 0 && threadIdx.x == 0) row[bins] = doc_off[d + 1] - doc_off[d];
}

// raw query code points -> dense codes of the corpus alphabet
__global__ __launch_bounds__(BLOCK) void query_map_kernel(const u32 *__restrict__ q_raw, u32 n_q,
                                                          const u32 *__restrict__ code_map,
                                                          const u32 *__restrict__ hi_bits,
                                                          const u32 *__restrict__ hi_rank, u32 hi_base,
                                                          u32 *__restrict__ q_code)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_q) return;
    const u32 c = q_raw[i];
    u32 code = Q_NOMATCH;
    if (c < EAST_HIP_TERMINATOR_START) {
        code = code_map[c];
        if (code == 0) code = Q_NOMATCH;          // symbol absent from the corpus
    } else if (hi_base && c < 0x110000u) {        // the corpus holds text at or above U+0A00 (tagged encoding): hi_base = its first code
        const u32 k = c - EAST_HIP_TERMINATOR_START;
        if ((hi_bits[k >> 5] >> (k & 31u)) & 1u)
            code = hi_base + hi_rank[k >> 5] + __popc(hi_bits[k >> 5] & ((1u << (k & 31u)) - 1u));
    }
    q_code[i] = code;
}

// The k-gram tables of a walk.  pairs == 0: one filled table of 4-byte entries over all k levels (see above).
// pairs != 0 (tables marked off the sorted window keys, window_sort.h: KgMark): kg3 = the filled table of the levels
// 1 .. k - 1, kg = the UNFILLED table of level k with 8-byte entries {rank, text position of the suffix at that rank}
// (0xFFFFFFFF: no suffix has this k-gram; entry [bins] = {n_d, -}) -- the end of a bucket is the rank of the next entry
// that is not empty (the one right behind it, for text that fills most of the table), and a walk that arrives in a
// bucket of ONE suffix goes on with that suffix's position out of the entry it has just read: one dependent gather
// (the symbol) instead of two (suffix array entry, then the symbol).
// The levels 1 .. k - 2 of a pair layout also exist as small tables of their own (`up`: per document the tables of A + 1,
// A^2 + 1, ... entries one after the other, copied out of kg3 once it is filled), so that EVERY level's interval is two
// adjacent entries -- one 8-byte load per level.  (The walk is bound by the number of address-divergent loads a
// wavefront issues -- every lane another sector --, not by bytes and not by the length of its dependency chain.)
#define WALK_ENDGAME 4u                    // suffixes a walk holds in registers once its interval is that small (score_walk_suffix)
#define KG_UP_LDS_WORDS 1024u              // the small upper tables of one document staged in LDS (A + 1 + A^2 + 1 <= 1024: A <= 30)
struct KgTables {
    const u32 *kg = nullptr, *kg3 = nullptr, *up = nullptr;
    int k = 0, pairs = 0;
    int up_lds = 0;                         // the walk kernel stages the document's `up` tables in LDS (they fit KG_UP_LDS_WORDS)
    int endgame = 1;                        // intervals of at most WALK_ENDGAME suffixes are finished out of registers (0: binary search to the end)
    u32 A = 0, bins = 0, up_stride = 0;
    // (set by the host, so that no walk divides: the bins of the table the levels 1 .. k3 read -- bins / A with pairs -- and
    // the stride of level i in it, A^(k3 - 1 - i))
    u32 bins3 = 0;
    u32 stride[KGRAM_KEYS_MAX_K] = {0, 0, 0, 0};
    void finish()
    {
        const int k3 = pairs ? k - 1 : k;
        bins3 = pairs ? bins / A : bins;
        u32 acc = 1;
        for (int i = KGRAM_KEYS_MAX_K - 1; i >= 0; i--)
            if (i < k3) { stride[i] = acc; acc *= A; }
    }
};
static_assert(KGRAM_KEYS_MAX_K == 4, "KgTables::stride");

// Two adjacent 4-byte entries (4-byte aligned) as ONE 8-byte load.  (A memcpy of 8 bytes is two 4-byte loads until the
// backend's vectoriser pairs them again, and it did not where the optimiser had first folded this path and the two-load
// path beside it into "two loads from two selected addresses": global_load_dword x 2 instead of dwordx2 in every walk --
// 1.27 against 1.18 ms for the walk of configs[2].  A typed load of an under-aligned 64-bit struct stays one load.)
struct __attribute__((packed, aligned(4))) U64Align4 { u64 v; };
struct __attribute__((packed, aligned(8))) U128Align8 { u64 lo, hi; };
struct __attribute__((packed, aligned(4))) U128Align4 { u32 a, b, c, d; };
__device__ __forceinline__ uint2 load_pair_u32(const u32 *p)
{
    const u64 v = reinterpret_cast<const U64Align4 *>(p)->v;
    return uint2{(u32)v, (u32)(v >> 32)};
}
__device__ __forceinline__ uint4 load_quad_u32(const void *p)      // 16 bytes, 8-byte aligned
{
    const U128Align8 v = *reinterpret_cast<const U128Align8 *>(p);
    return uint4{(u32)v.lo, (u32)(v.lo >> 32), (u32)v.hi, (u32)(v.hi >> 32)};
}

// The endgame of a walk: an interval of at most WALK_ENDGAME suffixes (behind four table levels a bucket of a 1 MiB
// document holds 2.4 on average).  A binary search would fetch suffix array entry and symbol of one rank after the other --
// 3 to 4 probes of two dependent loads each per query symbol; here the interval's entries come with ONE 16-byte load, the
// text behind every one of them four symbols at a time, side by side, and the walk goes on out of registers: the child
// interval of a symbol is the run of suffixes that carry it (symbols at a fixed depth are sorted inside an interval),
// counted.  Same intervals, same sums -- half the vector memory instructions per wavefront, which is what the walk is
// bound by, and two round trips instead of six to eight.  qsym(t): the query symbol at t.
// (Measured and not kept: only the members behind the first one fetched -- its position comes with the table entry --, with
// a load as wide as they are: two instructions where there was one, 1.12 -> 1.17 ms at configs[2].)
template <class SYM, class Q>
__device__ __forceinline__ void walk_endgame(const SYM *__restrict__ s, const u32 *__restrict__ sad, u32 nd, u32 root_ann, Q qsym,
                                             u32 t, u32 end, u32 lo, u32 hi, u32 &depth, u32 &nodes, double &acc, bool have_p,
                                             u32 p_lo, u32 &probes)
{
    const u32 n = hi - lo + 1;
    u32 p[WALK_ENDGAME];
    if (n == 1 && have_p) {
        p[0] = p_lo; p[1] = p[2] = p[3] = 0;
    } else if (nd >= WALK_ENDGAME) {
        const u32 base = lo + (WALK_ENDGAME - 1) < nd ? lo : nd - WALK_ENDGAME;     // (the 16 bytes stay inside the document's array)
        const U128Align4 q4 = *reinterpret_cast<const U128Align4 *>(sad + base);
        p[0] = q4.a; p[1] = q4.b; p[2] = q4.c; p[3] = q4.d;
        for (u32 sh = lo - base; sh > 0; sh--) { p[0] = p[1]; p[1] = p[2]; p[2] = p[3]; }
    } else {
#pragma unroll
        for (u32 i = 0; i < WALK_ENDGAME; i++) p[i] = i < n ? sad[lo + i] : 0u;
    }
    probes += n;
    u32 i0 = 0, i1 = n - 1;                               // the members of the interval still in the walk: p[i0 .. i1]
    u32 word[WALK_ENDGAME], d0 = depth;                   // their symbols at the depths d0 .. d0 + 3 (byte stream) / at d0
    auto fetch = [&]() {
#pragma unroll
        for (u32 i = 0; i < WALK_ENDGAME; i++) {
            const bool on = i >= i0 && i <= i1;
            if constexpr (sizeof(SYM) == 1) { u32 x = 0; if (on) __builtin_memcpy(&x, s + p[i] + depth, 4); word[i] = x; }
            else word[i] = on ? (u32)s[p[i] + depth] : 0u;
        }
        d0 = depth;
    };
    fetch();
    for (; t < end; t++) {
        const u32 c = qsym(t);
        if (c == Q_NOMATCH) break;
        if (depth - d0 >= (sizeof(SYM) == 1 ? 4u : 1u)) { fetch(); probes += (i1 - i0 + 2) >> 1; }
        const u32 shift = sizeof(SYM) == 1 ? 8u * (depth - d0) : 0u;
        u32 lt = 0, eq = 0;
#pragma unroll
        for (u32 i = 0; i < WALK_ENDGAME; i++) {
            const u32 x = sizeof(SYM) == 1 ? (word[i] >> shift) & 0xFFu : word[i];
            const bool on = i >= i0 && i <= i1;
            lt += on && x < c ? 1u : 0u;
            eq += on && x == c ? 1u : 0u;
        }
        if (eq == 0) break;                               // no suffix continues with c
        i0 += lt; i1 = i0 + eq - 1;
        const u32 a = lo + lt, b = a + eq - 1;
        if (b - a < hi - lo) {                            // (score_walk_suffix: enter)
            const u32 parent = depth == 0 ? root_ann : hi - lo + 1;
            acc += (double)(b - a + 1) / (double)parent;
            nodes++;
        }
        lo = a; hi = b; depth++;
    }
}

// One walk: the keyphrase suffix q_code[t0 .. end) down document d's annotated suffix array (sad / nd / root_ann).
// SYM = uint8_t: the byte stream (a quarter of the footprint; 0xFF terminators sort above every
// text code, which is all the binary search needs), SYM = u32: the dense symbol stream.
template <class SYM, class Q>
__device__ __forceinline__ double score_walk_suffix(const SYM *__restrict__ s, const u32 *__restrict__ sad, u32 nd, u32 root_ann,
                                                    Q q_sym, u32 t0, u32 end, int normalized,
                                                    const KgTables &kt, u32 d, u32 &probes, const u32 *up_lds = nullptr)
{
    u32 lo = 0, hi = nd - 1, depth = 0, nodes = 0;
    double acc = 0.0;
    u32 t = t0;
    u32 p_lo = 0;                       // the text position of the suffix at rank lo, where known
    bool have_p = false;
    auto enter = [&](u32 a, u32 b) {    // the interval of the next symbol; a node is entered whenever the interval shrinks
        if (b - a < hi - lo) {
            const u32 parent = depth == 0 ? root_ann : hi - lo + 1;
            acc += (double)(b - a + 1) / (double)parent;
            nodes++;
        }
        lo = a; hi = b; depth++;
    };
    if (kt.k > 0) {
        // The first k symbols: table reads instead of binary searches.  Where the entries lie follows from the query
        // symbols alone, so ALL of them -- every level's pair, the last level's entry with its successor -- are requested
        // before the first one is looked at: one memory round trip for the table levels instead of one per level (the
        // walk is bound by latency: a level's reads used to wait for the verdict of the level before).
        const int k3 = kt.pairs ? kt.k - 1 : kt.k;
        const u32 bins3 = kt.bins3;
        const u32 *row = (kt.pairs ? kt.kg3 : kt.kg) + (size_t)d * (bins3 + 1);
        const uint2 *row2 = reinterpret_cast<const uint2 *>(kt.kg) + (size_t)d * (kt.bins + 1);
        u32 cs[KGRAM_KEYS_MAX_K];
        int L = 0;                                        // query symbols the table levels can take
#pragma unroll
        for (int i = 0; i < KGRAM_KEYS_MAX_K; i++) {
            const u32 c = (i < kt.k && t0 + (u32)i < end) ? q_sym(t0 + (u32)i) : Q_NOMATCH;
            cs[i] = c;
            if (c != Q_NOMATCH && L == i) L = i + 1;
        }
        u32 ta[KGRAM_KEYS_MAX_K], tb[KGRAM_KEYS_MAX_K];
        u32 code = 0, up_off = 0, up_len = kt.A;
        const u32 *up = kt.up ? kt.up + (size_t)d * kt.up_stride : nullptr;
#pragma unroll
        for (int i = 0; i < KGRAM_KEYS_MAX_K; i++) {
            ta[i] = tb[i] = 0;
            if (i < k3 && i < L) {
                const u32 stride = kt.stride[i];
                code = code * kt.A + cs[i];
                if (stride != 1u && up_lds) {              // the document's small upper tables, staged by the workgroup
                    ta[i] = up_lds[up_off + code];
                    tb[i] = up_lds[up_off + code + 1u];
                } else if (stride == 1u || up) {           // two adjacent entries: one load
                    const uint2 ab = load_pair_u32(stride == 1u ? row + code : up + up_off + code);
                    ta[i] = ab.x; tb[i] = ab.y;
                } else {
                    ta[i] = row[code * stride];
                    tb[i] = row[(code + 1u) * stride];
                }
                up_off += up_len + 1u;
                up_len *= kt.A;
            }
        }
        uint2 e0 = uint2{0xFFFFFFFFu, 0u};
        u32 e1 = 0, g = 0;
        const bool last = kt.pairs && L == kt.k;           // (then k3 = k - 1 levels were taken above)
        if (last) {
            g = code * kt.A + cs[kt.k - 1];
            const uint4 ee = load_quad_u32(row2 + g);      // the entry and its successor: one 16-byte load (8-byte aligned)
            e0 = uint2{ee.x, ee.y};
            e1 = ee.z;
        }
        bool ended = false;
#pragma unroll
        for (int i = 0; i < KGRAM_KEYS_MAX_K; i++) {
            if (i < k3 && i < L && !ended) {
                probes += 2;
                if (tb[i] <= ta[i]) ended = true;          // no suffix continues with c
                else enter(ta[i], tb[i] - 1u);
            }
        }
        if (last && !ended) {
            probes += 2;
            if (e0.x == 0xFFFFFFFFu) ended = true;
            else {
                g++;
                while (e1 == 0xFFFFFFFFu) { e1 = row2[++g].x; probes++; }     // (entry [bins] is never empty)
                enter(e0.x, e1 - 1u);
                p_lo = e0.y; have_p = lo == hi;
            }
        }
        t = t0 + depth;
        if (ended || depth < (u32)kt.k) t = end;          // the walk ended inside the table levels
    }
    for (; t < end; t++) {
        if (kt.endgame && hi - lo < WALK_ENDGAME) break;                // few suffixes left: see below
        const u32 c = q_sym(t);
        if (c == Q_NOMATCH) break;
        u32 a, b;
        if (lo == hi) {
            probes++;
            if (!have_p) { p_lo = sad[lo]; have_p = true; }
            if (s[p_lo + depth] != c) break;
            a = b = lo;
        } else {
            u32 x = lo, y = hi + 1;                       // lower bound of c at this depth
            while (x < y) {
                const u32 mid = (x + y) >> 1;
                probes++;
                if (s[sad[mid] + depth] < c) x = mid + 1; else y = mid;
            }
            a = x;
            y = hi + 1;                                   // upper bound, from a
            while (x < y) {
                const u32 mid = (x + y) >> 1;
                probes++;
                if (s[sad[mid] + depth] <= c) x = mid + 1; else y = mid;
            }
            if (x == a) break;                            // no suffix continues with c
            b = x - 1;
        }
        enter(a, b);
    }
    if (kt.endgame && t < end && hi - lo < WALK_ENDGAME)
        walk_endgame<SYM>(s, sad, nd, root_ann, q_sym, t, end, lo, hi, depth, nodes, acc, have_p, p_lo, probes);
    double r = 0.0;
    if (depth > 0) {
        r = (acc + (double)depth) - (double)nodes;        // easa.py:127
        if (normalized) r /= (double)depth;               // easa.py:128-129
    }
    return r;
}

// One thread per (keyphrase suffix, document).  Two work layouts:
//   blk == nullptr  thread si of a document's workgroups takes suffix si; the results go to suffix_out (D_local x S) and
//                   score_reduce_kernel sums them;
//   blk != nullptr  a workgroup takes WHOLE keyphrases -- those numbered [blk[i], blk[i + 1]), at most BLOCK suffixes in
//                   all, packed by the host when the keyphrases were set -- and sums their suffix results itself, in
//                   suffix order, out of LDS: the same additions in the same order as the reduction kernel's, without
//                   the round trip of the per-suffix results through HBM and without the second launch.
template <class SYM>
__global__ __launch_bounds__(BLOCK) void score_walk_kernel(
    const SYM *__restrict__ s, const u32 *__restrict__ sa, const u32 *__restrict__ doc_off,
    const u32 *__restrict__ n_strings, u32 n_docs, const u32 *__restrict__ q_code,
    const u32 *__restrict__ q_end, u32 n_q, int normalized, KgTables kt, int xcd_order, u32 doc_first, u32 doc_count,
    double *__restrict__ suffix_out, unsigned long long *__restrict__ probe_count, const u32 *__restrict__ blk, u32 n_blk,
    const u32 *__restrict__ q_off, double *__restrict__ table)
{
    // XCD-aware work order: workgroups go round-robin over the 8 XCDs, each with its own 4 MB L2.  All the
    // keyphrase suffixes of ONE document are walked by ONE XCD (document d belongs to XCD d mod 8, which takes
    // its documents one after the other), so the top of that document's binary searches stays in that L2.
    // (Only with many documents -- xcd_order, host side: with a handful an XCD would sit idle.)
    __shared__ double res[BLOCK];
    __shared__ u32 up_stage[KG_UP_LDS_WORDS];
    const u32 blocks_per_doc = blk ? n_blk : (n_q + BLOCK - 1u) / BLOCK;
    const u32 local = xcd_order ? blockIdx.x >> 3 : blockIdx.x;
    // (the documents [doc_first, doc_first + doc_count) of this launch: the per-suffix scratch is bounded, see score_resident)
    const u32 dl = xcd_order ? (local / blocks_per_doc) * 8u + (blockIdx.x & 7u) : local / blocks_per_doc;
    if (dl >= doc_count) return;                          // (the whole workgroup)
    const u32 bi = local % blocks_per_doc;
    u32 kp0 = 0, kp1 = 0, s0 = bi * BLOCK, s1 = n_q;
    if (blk) { kp0 = blk[bi]; kp1 = blk[bi + 1]; s0 = q_off[kp0]; s1 = q_off[kp1]; }
    const u32 si = s0 + threadIdx.x;
    const u32 d = doc_first + dl;
    // (the thread's first query symbols and the end of its suffix are requested here, with the staging below: behind the
    // barrier they were two more round trips -- the end, then the symbols -- in front of the first table read)
    u32 q_pre[KGRAM_KEYS_MAX_K];
#pragma unroll
    for (u32 i = 0; i < KGRAM_KEYS_MAX_K; i++) q_pre[i] = q_code[si + i < n_q ? si + i : n_q - 1u];
    const u32 end_pre = q_end[si < n_q ? si : n_q - 1u];
    const u32 seg = doc_off[d], nd = doc_off[d + 1] - seg, root_ann = nd - n_strings[d];   // (uniform; in front of the barrier too)
    // The document's small upper tables (levels 1 .. k - 2 of a pair layout: A + 1, A^2 + 1 entries) come into LDS once per
    // workgroup, coalesced: every walk reads them, and the walk is bound by the number of requests its lanes send to the L2
    // (one per lane and level: each lane another line) -- two levels fewer of them.
    const u32 *up_lds = nullptr;
    if (kt.up_lds) {
        const u32 *src = kt.up + (size_t)d * kt.up_stride;
        // (all of a thread's words requested before the first is stored: a loop of unknown length is a round trip per step)
        static_assert(KG_UP_LDS_WORDS % BLOCK == 0, "whole rounds");
        u32 x[KG_UP_LDS_WORDS / BLOCK];
#pragma unroll
        for (u32 r = 0; r < KG_UP_LDS_WORDS / BLOCK; r++) {
            const u32 i = threadIdx.x + r * BLOCK;
            x[r] = src[i < kt.up_stride ? i : 0u];
        }
#pragma unroll
        for (u32 r = 0; r < KG_UP_LDS_WORDS / BLOCK; r++) {
            const u32 i = threadIdx.x + r * BLOCK;
            if (i < kt.up_stride) up_stage[i] = x[r];
        }
        __syncthreads();
        up_lds = up_stage;
    }
    double r = 0.0;
    if (si < s1) {
        u32 probes = 0;                 // table reads and binary-search probes of this walk (roofline accounting)
        auto q_sym = [&](u32 t) -> u32 {
            const u32 k = t - si;
            return k == 0 ? q_pre[0] : k == 1 ? q_pre[1] : k == 2 ? q_pre[2] : k == 3 ? q_pre[3] : q_code[t];
        };
        r = score_walk_suffix<SYM>(s, sa + seg, nd, root_ann, q_sym, si, end_pre, normalized, kt, d, probes, up_lds);
        if (suffix_out) suffix_out[(u64)dl * n_q + si] = r;
        if (probe_count) atomicAdd(probe_count, (unsigned long long)probes);   // (counting runs only: east_hip_score_probes)
    }
    if (!blk) return;
    res[threadIdx.x] = r;
    __syncthreads();
    if (threadIdx.x < kp1 - kp0) {
        const u32 k = kp0 + threadIdx.x;
        const u32 b = q_off[k] - s0, e = q_off[k + 1] - s0;
        double total = 0.0;
        for (u32 i = b; i < e; i++) total += res[i];      // easa.py:130, in suffix order
        table[(u64)k * n_docs + d] = total / (double)(e - b);   // easa.py:134
    }
}

// the small tables of the levels above kg3's own (KgTables::up), out of the filled kg3: level l entry g = kg3[g * A^(k3 - l)]
__global__ __launch_bounds__(BLOCK) void kgram_upper_kernel(const u32 *__restrict__ kg3, u32 bins3, u32 A, int k3, u32 up_stride,
                                                            u32 *__restrict__ up)
{
    const u32 d = blockIdx.x;
    const u32 *row = kg3 + (size_t)d * (bins3 + 1);
    u32 *out = up + (size_t)d * up_stride;
    u32 len = A, stride = bins3 / A;
    for (int l = 1; l < k3; l++) {
        for (u32 g = threadIdx.x; g <= len; g += BLOCK) out[g] = row[g * stride];
        out += len + 1u;
        len *= A;
        stride /= A;
    }
}

// the level-k table of a pair layout (KgTables): entry [bins] of every document = {n_d, -}
__global__ __launch_bounds__(BLOCK) void kgram_pairs_end_kernel(const u32 *__restrict__ doc_off, u32 n_docs, u32 bins,
                                                                u32 *__restrict__ kg)
{
    const u32 d = blockIdx.x * BLOCK + threadIdx.x;
    if (d >= n_docs) return;
    reinterpret_cast<uint2 *>(kg)[(size_t)d * (bins + 1) + bins] = uint2{doc_off[d + 1] - doc_off[d], 0u};
}

// out[k*D + d] = (sum of the keyphrase's suffix results, in suffix order) / |q|
__global__ __launch_bounds__(BLOCK) void score_reduce_kernel(const double *__restrict__ suffix,
                                                             const u32 *__restrict__ q_off,
                                                             u32 n_kp, u32 n_docs, u32 n_q, u32 doc_first, u32 doc_count,
                                                             double *__restrict__ out)
{
    // neighbouring threads take neighbouring keyphrases of ONE document: their suffix results lie side by
    // side in that document's row (the 8-byte stores into the K x D table are the strided side: 13x fewer)
    const u64 gid = (u64)blockIdx.x * BLOCK + threadIdx.x;
    if (gid >= (u64)n_kp * doc_count) return;
    const u32 d = (u32)(gid / n_kp);                      // (local to this launch's documents)
    const u32 k = (u32)(gid - (u64)d * n_kp);
    const u32 b = q_off[k], e = q_off[k + 1];
    const double *row = suffix + (u64)d * n_q;
    double total = 0.0;
    for (u32 i = b; i < e; i++) total += row[i];          // easa.py:130
    out[(u64)k * n_docs + doc_first + d] = total / (double)(e - b);   // easa.py:134
}

// Synonym-expanded scoring (easa.py:27-34): every keyphrase was expanded into its variants (the product of the
// per-word alternatives), all variants were scored as ordinary queries (table_v: V x D); the score of a keyphrase
// is the maximum over its variants -- a segmented max, variants of keyphrase k = [group_off[k], group_off[k+1]).
__global__ __launch_bounds__(BLOCK) void score_group_max_kernel(const double *__restrict__ table_v,
                                                                const u32 *__restrict__ group_off, u32 n_groups,
                                                                u32 n_docs, double *__restrict__ out)
{
    const u64 gid = (u64)blockIdx.x * BLOCK + threadIdx.x;
    if (gid >= (u64)n_groups * n_docs) return;
    const u32 k = (u32)(gid / n_docs), d = (u32)(gid - (u64)k * n_docs);
    const u32 b = group_off[k], e = group_off[k + 1];
    double best = table_v[(u64)b * n_docs + d];
    for (u32 v = b + 1; v < e; v++) {
        const double x = table_v[(u64)v * n_docs + d];
        best = x > best ? x : best;                         // (Python's max: the first maximal element; same value)
    }
    out[gid] = best;
}
