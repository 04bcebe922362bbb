// tables.h -- LCP table, annotation table and child tables of every document,
// computed from the (document-partitioned) suffix array.
//
//   lcptab  (easa.py:247-266, Kasai)  -> one thread per rank, direct comparison
//           of the two neighbouring suffixes (unique terminators end every
//           comparison, exactly as in the reference which has no bounds check)
//   anntab  (easa.py:306-331)         -> closed form over nearest smaller values
//           (SURVEY.md Appendix A.2): anntab[k] = NSV(k) - PSV(k) for the first
//           l-index of every lcp-interval, anntab[doc start] = n_d - m_d
//   childtab_next_l_index / up / down (easa.py:268-304) -> closed forms over
//           PSE / NSE and leftmost range minima
//
// Nearest-smaller-value and range-minimum queries run on a 16-ary min pyramid
// over lcptab (level i+1 = min of 16 entries = one 64-byte line of level i), so
// every query costs one group fetch (four independent 16-byte loads) per level,
// O(log16 n) worst case and one or two L2-resident groups in the common case;
// the sequential stacks of the reference disappear.
#pragma once
#include "common.h"

#define PYR_SHIFT 4
#define PYR_FAN 16u                 // one group = 16 words = one 64-byte line = 4 x 16-byte loads
#define PYR_MAX_LEVELS 9            // 16^8 > 2^31
#define NONE_U32 0xFFFFFFFFu

// Every level is padded to a multiple of 16 words with NONE_U32, so a whole
// group can always be fetched with four aligned 16-byte loads.
struct Pyramid {
    const u32 *ptr[PYR_MAX_LEVELS];
    u32 len[PYR_MAX_LEVELS];
    int levels;
};

static inline u32 pyr_padded(u32 len) { return (len + PYR_FAN - 1u) & ~(PYR_FAN - 1u); }
__device__ __forceinline__ u32 pyr_padded_dev(u32 len) { return (len + PYR_FAN - 1u) & ~(PYR_FAN - 1u); }

// document of a rank / position: last d with doc_off[d] <= x
__device__ __forceinline__ u32 doc_of(const u32 *__restrict__ doc_off, u32 n_docs, u32 x)
{
    u32 lo = 0, hi = n_docs;          // invariant: doc_off[lo] <= x < doc_off[hi]
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (doc_off[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

// coarse[i] = document of position i << shift; a position's document is then found by
// stepping forward from coarse[p >> shift] (almost always zero or one step)
__global__ __launch_bounds__(BLOCK) void doc_coarse_kernel(const u32 *__restrict__ doc_off, u32 n_docs, u32 n,
                                                           int shift, u32 n_coarse, u32 *__restrict__ coarse)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_coarse) return;
    const u64 p = (u64)i << shift;
    coarse[i] = doc_of(doc_off, n_docs, p < n ? (u32)p : n - 1u);
}

// keys[i] = document of the suffix sa[i]; four suffixes per thread (16-byte loads and stores; the
// arrays are padded to a multiple of 4 by the arena's 256-byte granularity)
__global__ __launch_bounds__(BLOCK) void doc_keys_kernel(const u32 *__restrict__ sa,
                                                         const u32 *__restrict__ doc_off,
                                                         const u32 *__restrict__ coarse, int shift,
                                                         u32 n, u32 *__restrict__ keys)
{
    const u32 i = (blockIdx.x * BLOCK + threadIdx.x) * 4u;
    if (i >= n) return;
    const uint4 p4 = *reinterpret_cast<const uint4 *>(sa + i);
    const u32 p[4] = {p4.x, p4.y, p4.z, p4.w};
    u32 d[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        d[e] = 0;
        if (i + e < n) {
            d[e] = coarse[p[e] >> shift];
            while (doc_off[d[e] + 1] <= p[e]) d[e]++;
        }
    }
    *reinterpret_cast<uint4 *>(keys + i) = uint4{d[0], d[1], d[2], d[3]};
}

// The same for up to DOC_LDS_MAX documents: the offsets live in LDS and every suffix is located by a
// branch-free binary search there (no gathers into the coarse table)
#define DOC_LDS_MAX 4096
__global__ __launch_bounds__(BLOCK) void doc_keys_lds_kernel(const u32 *__restrict__ sa,
                                                             const u32 *__restrict__ doc_off, u32 n_docs,
                                                             u32 n, u32 *__restrict__ keys)
{
    __shared__ u32 off[DOC_LDS_MAX + 1];
    for (u32 t = threadIdx.x; t <= n_docs; t += BLOCK) off[t] = doc_off[t];
    __syncthreads();
    const u32 i = (blockIdx.x * BLOCK + threadIdx.x) * 4u;
    if (i >= n) return;
    const uint4 p4 = *reinterpret_cast<const uint4 *>(sa + i);
    const u32 p[4] = {p4.x, p4.y, p4.z, p4.w};
    u32 lo[4] = {0, 0, 0, 0};                    // last d with off[d] <= p
    for (u32 step = 1u << (31 - __builtin_clz(n_docs)); step > 0; step >>= 1) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const u32 mid = lo[e] + step;
            if (mid < n_docs && off[mid] <= p[e]) lo[e] = mid;
        }
    }
    *reinterpret_cast<uint4 *>(keys + i) = uint4{lo[0], lo[1], lo[2], lo[3]};
}

// every document's table starts with 0 (its first rank has no left neighbour inside the document)
__global__ __launch_bounds__(BLOCK) void lcp_doc_starts_kernel(const u32 *__restrict__ doc_off, u32 n_docs,
                                                               u32 *__restrict__ lcp)
{
    const u32 d = blockIdx.x * BLOCK + threadIdx.x;
    if (d < n_docs) lcp[doc_off[d]] = 0;
}

__global__ __launch_bounds__(BLOCK) void lcp_kernel(const u32 *__restrict__ s,
                                                    const u32 *__restrict__ sa,
                                                    u32 n, u32 *__restrict__ lcp, u32 *__restrict__ capped, LcpBudget budget)
{
    const u32 r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) {
        if (r < ((n + PYR_FAN - 1u) & ~(PYR_FAN - 1u))) lcp[r] = NONE_U32;     // pyramid padding
        return;
    }
    if (r == 0) { lcp[0] = 0; return; }
    const u32 i = sa[r - 1], j = sa[r];
    u32 h = 0;
    // different documents never share a terminator, same document: both
    // suffixes end in distinct terminators => the comparison always stops in
    // bounds.  Four symbols per step (8 independent loads in flight); the three
    // pad words behind the stream make the look-ahead safe.
    bool deep = false, cut = false;
    while (true) {
        const u32 a0 = s[i + h], a1 = s[i + h + 1], a2 = s[i + h + 2], a3 = s[i + h + 3];
        const u32 b0 = s[j + h], b1 = s[j + h + 1], b2 = s[j + h + 2], b3 = s[j + h + 3];
        if (a0 != b0) break;
        if (a1 != b1) { h += 1; break; }
        if (a2 != b2) { h += 2; break; }
        if (a3 != b3) { h += 3; break; }
        h += 4;
        if (h >= LCP_SOFT_CAP && !deep) {                    // (see LCP_SOFT_CAP, common.h)
            if (!lcp_deep_allowed(budget)) { cut = true; break; }
            deep = true;
        }
        if (h >= LCP_DIRECT_CAP) { cut = true; break; }
    }
    // (the first rank of every document is reset to 0 by lcp_doc_starts_kernel)
    if (cut) { h |= LCP_PARTIAL_BIT; raise_flag(capped); }
    lcp[r] = h;
}

// LCP for small text alphabets (sigma_text <= 254): the symbol stream is also
// kept as one byte per symbol (text code, 0xFF = "a terminator"), a quarter of
// the footprint, so the two random windows per rank mostly come out of the
// Infinity Cache.  Eight symbols per step: first differing byte by xor + ctz;
// equal 0xFF bytes are two DIFFERENT terminators, so a 0xFF byte also ends the
// common prefix.  The stream is padded with 16 zero bytes.
__global__ __launch_bounds__(BLOCK) void lcp8_kernel(const uint8_t *__restrict__ s8,
                                                     const u32 *__restrict__ sa, u32 n,
                                                     u32 *__restrict__ lcp, u32 *__restrict__ capped, LcpBudget budget)
{
    const u32 r = blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = r < n;
    // every rank fetches the 8-symbol window of ITS suffix once; the window of the left
    // neighbour comes from the lane below (only lane 0 of a wave gathers twice)
    const u32 j = valid ? sa[r] : 0u;
    const u64 b = load_u64_unaligned(s8 + j);
    u32 i = __shfl_up(j, 1, WAVE);
    u32 a_lo = __shfl_up((u32)b, 1, WAVE), a_hi = __shfl_up((u32)(b >> 32), 1, WAVE);
    u64 a = ((u64)a_hi << 32) | a_lo;
    if (lane_id() == 0 && valid && r > 0) {
        i = sa[r - 1];
        a = load_u64_unaligned(s8 + i);
    }
    if (!valid) {
        if (r < ((n + PYR_FAN - 1u) & ~(PYR_FAN - 1u))) lcp[r] = NONE_U32;     // pyramid padding
        return;
    }
    if (r == 0) { lcp[0] = 0; return; }
    u32 h;
    {
        const u64 d = a ^ b, z = ~a;                                 // zero byte of z <=> 0xFF in a
        const u64 t = (z - 0x0101010101010101ull) & ~z & 0x8080808080808080ull;
        const u32 mism = d ? (u32)__builtin_ctzll(d) >> 3 : 8u;
        const u32 term = t ? (u32)__builtin_ctzll(t) >> 3 : 8u;
        h = mism < term ? mism : term;
    }
    if (h == 8u) h = lcp_bytes_capped(s8, i, j, 8u, budget);         // (most ranks end inside the first windows)
    // (the first rank of every document is reset to 0 by lcp_doc_starts_kernel)
    if (h & LCP_PARTIAL_BIT) raise_flag(capped);
    lcp[r] = h;
}

// ---- finishing pass for unfinished LCP entries (LCP_PARTIAL_BIT) ----------------------------------------------------
// rank[p] = rank of the suffix at p; PLCP(p) = lcp[rank[p]]; phi(p) = sa[rank[p] - 1], the suffix in front of p's.
// Kasai's bound PLCP(p) >= PLCP(p - 1) - 1 lets a walk over the text resume every comparison where the one before ended,
// but a walk has to START somewhere, and a start inside a long repeat compares its whole length: a blocked walk costs
// (n / block) x the mean LCP -- quadratic on a run of one symbol, a short period, a passage written twice (measured, 32
// positions per thread: 144 of 162 ms for 4 M symbols of one letter, 0.8 s for 16 M).  What is used instead is the
// irreducible-LCP lemma (Karkkainen, Manzini, Puglisi: Permuted Longest-Common-Prefix Array, CPM 2009): if the symbols
// IN FRONT of p and of phi(p) agree (p is "reducible"), PLCP(p) = PLCP(p - 1) - 1 exactly -- no comparison at all --, and
// the PLCP values of the irreducible positions sum to at most 2 n log n.  So:
//   classify  one thread per position: finished entries and irreducible positions are ANCHORS (the latter are listed);
//   compare   the listed positions get their entry by comparing text, a wavefront each, 512 symbols per step (entries
//             longer than 8 K symbols: a workgroup of 1024 each, 32 K symbols per step);
//   fill      every other position p: PLCP(p) = PLCP(j) - (p - j), j the nearest anchor in front of p (found from an
//             inclusive count of the anchors and their compacted positions).
// All four passes are data-parallel; no text is read for a reducible position.  (A document's first position is
// irreducible -- the symbol in front of it is the terminator of the document before, which equals nothing --, so no run
// crosses a document.)  BYTES: byte stream with the 0xFF rule (equal 0xFF bytes are different terminators), else exact
// u32 symbols, whose terminators are numbered apart.
__global__ __launch_bounds__(BLOCK) void inverse_sa_kernel(const u32 *__restrict__ sa, u32 n, u32 *__restrict__ rank)
{
    const u32 r = blockIdx.x * BLOCK + threadIdx.x;
    if (r < n) rank[sa[r]] = r;
}

struct U8In {                                     // one flag per element, a byte each
    const uint8_t *p;
    __device__ __forceinline__ u32 operator()(u32 i) const { return p[i]; }
};

template <bool BYTES>
__global__ __launch_bounds__(BLOCK) void lcp_phi_classify_kernel(const void *__restrict__ sym, const u32 *__restrict__ sa,
                                                                 const u32 *__restrict__ rank, const u32 *__restrict__ lcp, u32 n,
                                                                 uint8_t *__restrict__ anchor, u32 *__restrict__ list,
                                                                 u32 *__restrict__ list_n)
{
    const u32 p = blockIdx.x * BLOCK + threadIdx.x;
    bool irreducible = false;
    if (p < n) {
        const u32 r = rank[p];
        const u32 h = lcp[r];
        bool is_anchor = true;
        if (h & LCP_PARTIAL_BIT) {                 // (never a document's first rank: those hold 0)
            const u32 q = sa[r - 1];
            bool reducible = false;
            if (p > 0 && q > 0) {
                if (BYTES) {
                    const uint8_t a = ((const uint8_t *)sym)[p - 1], b = ((const uint8_t *)sym)[q - 1];
                    reducible = a == b && a != 0xFFu;
                } else {
                    reducible = ((const u32 *)sym)[p - 1] == ((const u32 *)sym)[q - 1];
                }
            }
            is_anchor = irreducible = !reducible;
        }
        anchor[p] = is_anchor ? 1 : 0;
    }
    const u64 bal = __ballot(irreducible);         // one list append per wavefront
    if (bal) {
        u32 at = 0;
        if (lane_id() == (u32)__builtin_ctzll(bal)) at = atomicAdd(list_n, (u32)__popcll(bal));
        at = __shfl(at, __builtin_ctzll(bal), WAVE) + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
        if (irreducible) list[at] = p;
    }
}

// one step of a cooperative comparison: every lane looks at 8 symbols (BYTES) or 1 symbol at its own offset; returns
// the number of symbols of the step's stretch that agree before the first difference (or terminator), stretch if all do
template <bool BYTES>
__device__ __forceinline__ u32 lcp_phi_step(const void *__restrict__ sym, u32 p, u32 q, u32 at, u32 lanes_rank, u32 &stretch)
{
    if (BYTES) {
        const uint8_t *s8 = (const uint8_t *)sym;
        const u64 x = load_u64_unaligned(s8 + p + at + 8u * lanes_rank), y = load_u64_unaligned(s8 + q + at + 8u * lanes_rank);
        const u64 d = x ^ y, z = ~x;
        const u64 t = (z - 0x0101010101010101ull) & ~z & 0x8080808080808080ull;
        const u32 mism = d ? (u32)__builtin_ctzll(d) >> 3 : 8u;
        const u32 term = t ? (u32)__builtin_ctzll(t) >> 3 : 8u;
        stretch = 8u;
        return mism < term ? mism : term;
    } else {
        const u32 *s = (const u32 *)sym;
        stretch = 1u;
        return s[p + at + lanes_rank] == s[q + at + lanes_rank] ? 1u : 0u;
    }
}

// the listed positions, a wavefront each; what is not decided within PHI_WAVE_STEPS steps goes to the long list
#define PHI_WAVE_STEPS 16u
template <bool BYTES>
__global__ __launch_bounds__(BLOCK) void lcp_phi_compare_kernel(const void *__restrict__ sym, const u32 *__restrict__ sa,
                                                                const u32 *__restrict__ rank, const u32 *__restrict__ list,
                                                                const u32 *__restrict__ list_n, u32 n, u32 *__restrict__ lcp,
                                                                u32 *__restrict__ long_list, u32 *__restrict__ long_n)
{
    const u32 count = *list_n;
    for (u32 i = blockIdx.x * WAVES_PER_BLOCK + wave_id(); i < count; i += gridDim.x * WAVES_PER_BLOCK) {
        const u32 p = list[i], r = rank[p], q = sa[r - 1];
        u32 h = lcp[r] & ~LCP_PARTIAL_BIT;         // what the direct comparison had seen to agree when it was cut
        // (the stream is readable far enough behind every position a comparison can reach: it ends at the document's last
        // terminator at the latest, a step looks at most 512 bytes / 64 words past that -- inside the arena, see build_impl)
        const u32 limit = n - (p > q ? p : q);      // symbols left behind the later of the two positions
        bool done = false;
        for (u32 step = 0; step < PHI_WAVE_STEPS && !done; step++) {
            u32 stretch;
            const u32 agree = lcp_phi_step<BYTES>(sym, p, q, h, lane_id(), stretch);
            const u64 bal = __ballot(agree < stretch);
            if (bal) {
                const u32 first = (u32)__builtin_ctzll(bal);
                h += first * stretch + __shfl(agree, first, WAVE);
                done = true;
            } else {
                h += WAVE * stretch;
                if (h >= limit) done = true;        // (cannot happen: the last symbol is a terminator of its own)
            }
        }
        if (lane_id() == 0) {
            if (done) lcp[r] = h < limit ? h : limit;
            else {
                lcp[r] = LCP_PARTIAL_BIT | h;
                long_list[atomicAdd(long_n, 1u)] = p;
            }
        }
    }
}

// the long list: a workgroup of PHI_LONG_THREADS per position
#define PHI_LONG_THREADS 1024
template <bool BYTES>
__global__ __launch_bounds__(PHI_LONG_THREADS) void lcp_phi_long_kernel(const void *__restrict__ sym, const u32 *__restrict__ sa,
                                                                        const u32 *__restrict__ rank, const u32 *__restrict__ list,
                                                                        const u32 *__restrict__ list_n, u32 n, u32 *__restrict__ lcp)
{
    __shared__ u32 first_wave[PHI_LONG_THREADS / WAVE];
    for (u32 i = blockIdx.x; i < *list_n; i += gridDim.x) {
        const u32 p = list[i], r = rank[p], q = sa[r - 1];
        u32 h = lcp[r] & ~LCP_PARTIAL_BIT;
        const u32 limit = n - (p > q ? p : q);
        while (true) {
            u32 stretch;
            const u32 agree = lcp_phi_step<BYTES>(sym, p, q, h, threadIdx.x, stretch);
            const u64 bal = __ballot(agree < stretch);
            u32 mine = 0xFFFFFFFFu;                 // symbols of the step's stretch in front of this wave's first difference
            if (bal) {
                const u32 first = (u32)__builtin_ctzll(bal);
                mine = (wave_id() * WAVE + first) * stretch + __shfl(agree, first, WAVE);
            }
            if (lane_id() == 0) first_wave[wave_id()] = mine;
            __syncthreads();
            u32 best = 0xFFFFFFFFu;
            for (u32 w = 0; w < PHI_LONG_THREADS / WAVE; w++) best = first_wave[w] < best ? first_wave[w] : best;
            __syncthreads();
            if (best != 0xFFFFFFFFu) { h += best; break; }
            h += PHI_LONG_THREADS * stretch;
            if (h >= limit) break;
        }
        if (threadIdx.x == 0) lcp[r] = h < limit ? h : limit;
    }
}

// the anchors' positions, compacted: apos[c - 1] = p for anchor p with inclusive count c
__global__ __launch_bounds__(BLOCK) void lcp_phi_anchors_kernel(const uint8_t *__restrict__ anchor, const u32 *__restrict__ count, u32 n,
                                                                u32 *__restrict__ apos)
{
    const u32 p = blockIdx.x * BLOCK + threadIdx.x;
    if (p < n && anchor[p]) apos[count[p] - 1u] = p;
}

// every position that is no anchor: its entry from the nearest anchor in front of it
__global__ __launch_bounds__(BLOCK) void lcp_phi_fill_kernel(const uint8_t *__restrict__ anchor, const u32 *__restrict__ count,
                                                             const u32 *__restrict__ apos, const u32 *__restrict__ rank, u32 n,
                                                             u32 *__restrict__ lcp)
{
    const u32 p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= n || anchor[p]) return;
    const u32 j = apos[count[p] - 1u];             // (position 0 is an anchor: the count is at least 1)
    lcp[rank[p]] = lcp[rank[j]] - (p - j);
}

// out[i] = min of in[16i .. 16i+15]; the padding of out becomes NONE_U32
__global__ __launch_bounds__(BLOCK) void pyramid_level_kernel(const u32 *__restrict__ in, u32 len_out,
                                                              u32 len_out_padded, u32 *__restrict__ out)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= len_out_padded) return;
    u32 v = NONE_U32;
    if (i < len_out) {
        const uint4 *g = reinterpret_cast<const uint4 *>(in + ((size_t)i << PYR_SHIFT));
        const uint4 a = g[0], b = g[1], c = g[2], d = g[3];
        const u32 m0 = min(min(a.x, a.y), min(a.z, a.w)), m1 = min(min(b.x, b.y), min(b.z, b.w));
        const u32 m2 = min(min(c.x, c.y), min(c.z, c.w)), m3 = min(min(d.x, d.y), min(d.z, d.w));
        v = min(min(m0, m1), min(m2, m3));
    }
    out[i] = v;
}

// The top of the pyramid in one launch: level `first` (at most PYR_TOP entries) from the level below, and every level
// above it from LDS.  One workgroup; len[] / ptr[]: as in Pyramid.
#define PYR_TOP 4096u
__global__ __launch_bounds__(BLOCK) void pyramid_top_kernel(Pyramid P, int first)
{
    __shared__ u32 cur[PYR_TOP], nxt[PYR_TOP / PYR_FAN];
    const u32 *below = P.ptr[first - 1];
    u32 len = P.len[first];
    for (u32 i = threadIdx.x; i < pyr_padded_dev(len); i += BLOCK) {
        u32 v = NONE_U32;
        if (i < len) {
            const uint4 *g = reinterpret_cast<const uint4 *>(below + ((size_t)i << PYR_SHIFT));
            const uint4 a = g[0], b = g[1], c = g[2], d = g[3];
            v = min(min(min(a.x, a.y), min(a.z, a.w)), min(min(b.x, b.y), min(b.z, b.w)));
            v = min(v, min(min(min(c.x, c.y), min(c.z, c.w)), min(min(d.x, d.y), min(d.z, d.w))));
        }
        cur[i] = v;
        const_cast<u32 *>(P.ptr[first])[i] = v;
    }
    __syncthreads();
    u32 *a = cur, *b = nxt;
    for (int lvl = first + 1; lvl < P.levels; lvl++) {
        const u32 len_out = P.len[lvl];
        for (u32 i = threadIdx.x; i < pyr_padded_dev(len_out); i += BLOCK) {
            u32 v = NONE_U32;
            if (i < len_out)
                for (u32 q = 0; q < PYR_FAN; q++) v = min(v, a[i * PYR_FAN + q]);   // (the padding of the level below holds NONE)
            b[i] = v;
            const_cast<u32 *>(P.ptr[lvl])[i] = v;
        }
        __syncthreads();
        u32 *t = a; a = b; b = t;
    }
}

// bit i set iff M[start + i] < v (STRICT) or <= v, for the 16-word group at `start`
template <bool STRICT>
__device__ __forceinline__ u32 pyr_group_mask(const u32 *__restrict__ M, u32 start, u32 v)
{
    const uint4 *g = reinterpret_cast<const uint4 *>(M + start);
    const uint4 a = g[0], b = g[1], c = g[2], d = g[3];
    const u32 x[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    u32 m = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) m |= (STRICT ? x[i] < v : x[i] <= v) ? (1u << i) : 0u;
    return m;
}

// largest p < k with level0[p] < v (STRICT) or <= v; NONE_U32 if none.
// One group (four independent 16-byte loads) per level on the way up and down.
template <bool STRICT>
__device__ __forceinline__ u32 pyr_find_left(const Pyramid &P, u32 k, u32 v)
{
    u32 pos = k;
    int lvl = 0;
    while (true) {
        const u32 start = pos & ~(PYR_FAN - 1u);
        if (pos > start) {
            const u32 m = pyr_group_mask<STRICT>(P.ptr[lvl], start, v) & ((1u << (pos - start)) - 1u);
            if (m) {
                u32 q = start + 31u - (u32)__clz((int)m);
                while (lvl > 0) {
                    lvl--;
                    const u32 base = q << PYR_SHIFT;
                    const u32 mm = pyr_group_mask<STRICT>(P.ptr[lvl], base, v);
                    q = base + 31u - (u32)__clz((int)mm);
                }
                return q;
            }
        }
        if (start == 0 || lvl + 1 >= P.levels) return NONE_U32;
        pos = start >> PYR_SHIFT;
        lvl++;
    }
}

// smallest q > k with level0[q] < v (STRICT) or <= v; NONE_U32 if none
template <bool STRICT>
__device__ __forceinline__ u32 pyr_find_right(const Pyramid &P, u32 k, u32 v)
{
    u32 pos = k + 1;
    int lvl = 0;
    while (true) {
        const u32 len = P.len[lvl];
        if (pos >= len) return NONE_U32;
        const u32 start = pos & ~(PYR_FAN - 1u);
        const u32 m = pyr_group_mask<STRICT>(P.ptr[lvl], start, v) & ~((1u << (pos - start)) - 1u);
        if (m) {
            u32 q = start + (u32)__ffs((int)m) - 1u;
            while (lvl > 0) {
                lvl--;
                const u32 base = q << PYR_SHIFT;
                const u32 mm = pyr_group_mask<STRICT>(P.ptr[lvl], base, v);
                q = base + (u32)__ffs((int)mm) - 1u;
            }
            return q;
        }
        const u32 next = start + PYR_FAN;
        if (next >= len || lvl + 1 >= P.levels) return NONE_U32;
        pos = next >> PYR_SHIFT;
        lvl++;
    }
}

// pyr_find_left<false> and pyr_find_right<true> of one rank at once, level by level: first both searches
// climb together (one group each per level, the two fetches in flight side by side), then both descend
// together.  Written for wavefronts whose lanes search intervals of very different widths: with the two
// plain searches above the lanes that find their group at different levels run their descents one after
// the other (SIMT), 40-60 dependent fetches per wavefront instead of about a dozen.
__device__ __forceinline__ void pyr_find_both(const Pyramid &P, u32 k, u32 v, u32 &pse, u32 &nsv)
{
    u32 lpos = k, rpos = k + 1, lq = NONE_U32, rq = NONE_U32;
    int llvl = -1, rlvl = -1;                       // level at which the search found its group (-1: nothing found)
    bool ldone = false, rdone = false;
    for (int lvl = 0; lvl < P.levels && !(ldone && rdone); lvl++) {
        const u32 *M = P.ptr[lvl];
        if (!ldone) {
            const u32 start = lpos & ~(PYR_FAN - 1u);
            u32 m = 0;
            if (lpos > start) m = pyr_group_mask<false>(M, start, v) & ((1u << (lpos - start)) - 1u);
            if (m) { lq = start + 31u - (u32)__clz((int)m); llvl = lvl; ldone = true; }
            else if (start == 0) ldone = true;
            else lpos = start >> PYR_SHIFT;
        }
        if (!rdone) {
            const u32 len = P.len[lvl];
            if (rpos >= len) rdone = true;
            else {
                const u32 start = rpos & ~(PYR_FAN - 1u);
                const u32 m = pyr_group_mask<true>(M, start, v) & ~((1u << (rpos - start)) - 1u);
                if (m) { rq = start + (u32)__ffs((int)m) - 1u; rlvl = lvl; rdone = true; }
                else if (start + PYR_FAN >= len) rdone = true;
                else rpos = (start + PYR_FAN) >> PYR_SHIFT;
            }
        }
    }
    for (int lvl = P.levels - 1; lvl > 0; lvl--) {
        const u32 *M = P.ptr[lvl - 1];
        if (llvl == lvl) {
            const u32 base = lq << PYR_SHIFT;
            lq = base + 31u - (u32)__clz((int)pyr_group_mask<false>(M, base, v));
            llvl--;
        }
        if (rlvl == lvl) {
            const u32 base = rq << PYR_SHIFT;
            rq = base + (u32)__ffs((int)pyr_group_mask<true>(M, base, v)) - 1u;
            rlvl--;
        }
    }
    pse = lq;
    nsv = rq;
}

// leftmost position of the minimum of level0 over the open interval (a, b),
// a + 1 < b.  Walks whole groups through the pyramid.
__device__ __forceinline__ u32 pyr_leftmost_argmin(const Pyramid &P, u32 a, u32 b)
{
    // 1. minimum value over (a, b)
    const u32 lo = a + 1;
    u32 best = NONE_U32;
    {
        u32 l = lo, h = b;            // [l, h)
        int lvl = 0;
        while (l < h) {
            const u32 *M = P.ptr[lvl];
            // peel unaligned heads/tails at this level, then go up
            while (l < h && (l & (PYR_FAN - 1u))) { const u32 x = M[l]; best = x < best ? x : best; l++; }
            while (l < h && (h & (PYR_FAN - 1u))) { h--; const u32 x = M[h]; best = x < best ? x : best; }
            if (l >= h) break;
            if (lvl + 1 >= P.levels) {
                for (u32 q = l; q < h; q++) { const u32 x = M[q]; best = x < best ? x : best; }
                break;
            }
            l >>= PYR_SHIFT; h >>= PYR_SHIFT; lvl++;
        }
    }
    // 2. first position >= lo holding a value <= best (it is == best and < b)
    if (P.ptr[0][lo] == best) return lo;
    return pyr_find_right<false>(P, lo, best);
}

// anntab (easa.py:306-331) in two launches.
//
// ann_stream_kernel: a workgroup stages 1024 consecutive LCP values plus ANN_HALO to either side in LDS.
//   phase 1  every thread decides its 4 ranks from the 8 neighbours to either side (five 16-byte LDS
//            reads); most ranks end here and the thread writes one 16-byte store;
//   phase 2  the rest -- first l-indices of intervals wider than that, a few per cent -- go to an LDS
//            work list and are looked at by eight lanes each over the staged values, at most ANN_LOCAL ranks
//            to either side (one rank per thread: the tile's few dozen ranks queue up in one wavefront; a wave
//            per rank with 64 neighbours per ballot was measured: twice as slow, the ranks then queue up
//            behind one another);
//   what is wider still (the top levels of the tree, about one rank in a hundred) is appended to a
//   global list for ann_wide_kernel.
//   The kernel also writes level 1 of the min pyramid (it has the values in LDS): the LCP table is read
//   once for both.
// ann_wide_kernel: the listed ranks, one per thread, through the pyramid (O(log16 n) groups each).
// Document starts carry lcp == 0, which bounds every search inside the document; ranks outside
// [0, n) read as 0.
#ifndef ANN_NEAR
#define ANN_NEAR 8
#endif
#define ANN_IPT 4                       // consecutive ranks per thread: 16-byte loads and stores
#define ANN_TILE (BLOCK * ANN_IPT)
// (round 4, with eight lanes per rank in phase 2 -- halo / walk 32 / 24, 64 / 64, 96 / 96, 128 / 128: configs[2] ann_stream +
// ann_wide 0.77 + 0.63, 0.84 + 0.13, 0.93 + 0.13, 1.03 + 0.13 ms -- its depth-3 intervals are 51 ranks wide --; configs[1]
// 0.16 + 0.08, 0.17 + 0.08, 0.18 + 0.07, 0.20 + 0.04; the Zipf stand-in 0.32 + 0.14, 0.37 + 0.05, ...)
#ifndef ANN_HALO
#define ANN_HALO 64
#endif
#ifndef ANN_LOCAL
#define ANN_LOCAL 64                    // phase 2 looks at most this far; what is wider goes to ann_wide_kernel
#endif

__global__ __launch_bounds__(BLOCK) void ann_stream_kernel(const u32 *__restrict__ lcp, const u32 *__restrict__ doc_off,
                                                           const u32 *__restrict__ n_strings, u32 n_docs, u32 n,
                                                           u32 *__restrict__ ann, u32 *__restrict__ lvl1, u32 len1,
                                                           u32 len1_padded, u32 *__restrict__ wide_list,
                                                           u32 *__restrict__ wide_count,      // per tile: its own stretch / count
                                                           u32 *__restrict__ lcp_pad)         // != nullptr: the table's padding is still to be written
{
    __shared__ __attribute__((aligned(16))) u32 tile[ANN_TILE + 2 * ANN_HALO];
    __shared__ u32 work[ANN_TILE];
    __shared__ u32 work_count, far_count;
    if (threadIdx.x == 0) { work_count = 0; far_count = 0; }
    // (the entries between n and the next multiple of 16 -- the pyramid searches of ann_wide_kernel read whole groups;
    // nothing in this kernel depends on them: ranks outside [0, n) are masked below)
    if (lcp_pad && blockIdx.x == gridDim.x - 1 && threadIdx.x < PYR_FAN && n + threadIdx.x < ((n + PYR_FAN - 1u) & ~(PYR_FAN - 1u)))
        lcp_pad[n + threadIdx.x] = NONE_U32;
    const u32 tile_base = blockIdx.x * ANN_TILE;
    const u32 k0 = tile_base + threadIdx.x * ANN_IPT;
    {
        // own four values (the table is padded to a multiple of 16 entries, so whole groups can be loaded up
        // to there); ranks outside [0, n) read as 0, which stops every scan
        uint4 x = {0u, 0u, 0u, 0u};
        if (k0 < ((n + PYR_FAN - 1u) & ~(PYR_FAN - 1u))) x = *reinterpret_cast<const uint4 *>(lcp + k0);
        if (k0 + 0 >= n) x.x = 0u;
        if (k0 + 1 >= n) x.y = 0u;
        if (k0 + 2 >= n) x.z = 0u;
        if (k0 + 3 >= n) x.w = 0u;
        // (the halo is requested before the stretch is staged: one round trip for both)
        const bool has_halo = threadIdx.x < 2 * ANN_HALO / 4;       // ANN_HALO ranks to the left (first threads) and to the right (the next ones)
        const bool left = threadIdx.x < ANN_HALO / 4;
        const u32 q = left ? threadIdx.x : threadIdx.x - ANN_HALO / 4;
        const i64 g = left ? (i64)tile_base - ANN_HALO + 4 * q : (i64)tile_base + ANN_TILE + 4 * q;
        uint4 h = {0u, 0u, 0u, 0u};
        if (has_halo && g >= 0 && g < (i64)((n + PYR_FAN - 1u) & ~(PYR_FAN - 1u))) h = *reinterpret_cast<const uint4 *>(lcp + g);
        *reinterpret_cast<uint4 *>(&tile[ANN_HALO + threadIdx.x * ANN_IPT]) = x;
        if (has_halo) {
            if (g + 0 >= (i64)n) h.x = 0u;
            if (g + 1 >= (i64)n) h.y = 0u;
            if (g + 2 >= (i64)n) h.z = 0u;
            if (g + 3 >= (i64)n) h.w = 0u;
            *reinterpret_cast<uint4 *>(&tile[left ? 4 * q : ANN_HALO + ANN_TILE + 4 * q]) = h;
        }
    }
    __syncthreads();
    // level 1 of the min pyramid: entry e = min of the 16 ranks 16e .. 16e+15 (entries past the table: NONE)
    if (threadIdx.x < ANN_TILE / PYR_FAN) {
        const u32 e = tile_base / PYR_FAN + threadIdx.x;
        if (e < len1_padded) {
            u32 m = NONE_U32;
            if (e < len1) {
                const uint4 *g = reinterpret_cast<const uint4 *>(&tile[ANN_HALO + threadIdx.x * PYR_FAN]);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint4 y = g[q];
                    const u32 r = e * PYR_FAN + 4u * q;          // (ranks past n hold 0 in the tile: not part of the minimum)
                    if (r + 0 < n) m = min(m, y.x);
                    if (r + 1 < n) m = min(m, y.y);
                    if (r + 2 < n) m = min(m, y.z);
                    if (r + 3 < n) m = min(m, y.w);
                }
            }
            lvl1[e] = m;
        }
    }
    if (k0 < n) {
        // c[i] = lcp[k0 - ANN_NEAR + i]: the thread's 4 ranks and ANN_NEAR ranks to either side
        u32 c[ANN_IPT + 2 * ANN_NEAR];
#pragma unroll
        for (int q = 0; q < (ANN_IPT + 2 * ANN_NEAR) / 4; q++) {
            const uint4 x = *reinterpret_cast<const uint4 *>(&tile[ANN_HALO + threadIdx.x * ANN_IPT - ANN_NEAR + 4 * q]);
            c[4 * q] = x.x; c[4 * q + 1] = x.y; c[4 * q + 2] = x.z; c[4 * q + 3] = x.w;
        }
        u32 out[ANN_IPT];
        bool direct[ANN_IPT];
#pragma unroll
        for (int e = 0; e < ANN_IPT; e++) {
            const u32 k = k0 + e;
            const int i = ANN_NEAR + e;                  // c[i] = lcp[k]
            const u32 v = c[i];
            out[e] = 0;
            direct[e] = true;
            if (k >= n) continue;
            if (v == 0) {
                // only a document's first rank carries an annotation here: the root, n_d - m_d
                const u32 d = n_docs > 1 ? doc_of(doc_off, n_docs, k) : 0u;
                out[e] = doc_off[d] == k ? (doc_off[d + 1] - k) - n_strings[d] : 0u;
                continue;
            }
            // nearest previous value <= v within ANN_NEAR ranks (lcp[0] == 0 and the zero halo stop the scan)
            u32 back = 0, x = 0;
#pragma unroll
            for (int t = ANN_NEAR; t >= 1; t--)
                if (c[i - t] <= v) { back = (u32)t; x = c[i - t]; }
            if (back == 0) { direct[e] = false; continue; }      // further away: phase 2
            if (x < v) {                        // first l-index of its interval: width = NSV - PSV
                u32 fwd = 0;
#pragma unroll
                for (int t = ANN_NEAR; t >= 1; t--)
                    if (c[i + t] < v) fwd = (u32)t;
                if (fwd == 0) direct[e] = false;
                else out[e] = fwd + back;
            }
        }
        if (direct[0] && direct[1] && direct[2] && direct[3] && k0 + ANN_IPT <= n) {
            *reinterpret_cast<uint4 *>(ann + k0) = uint4{out[0], out[1], out[2], out[3]};
        } else {
#pragma unroll
            for (int e = 0; e < ANN_IPT; e++) {
                if (k0 + e >= n) break;
                if (direct[e]) ann[k0 + e] = out[e];
                else work[atomicAdd(&work_count, 1u)] = threadIdx.x * ANN_IPT + e;
            }
        }
    }
    __syncthreads();
    // phase 2: EIGHT lanes per rank -- lane q looks at the distances q + 1, q + 9, q + 17, ... to either side, the nearest hit
    // of the eight comes from three shuffles.  (One rank per thread left a tile's two or three dozen ranks to the first
    // lanes of ONE wavefront, each walking up to 2 x ANN_LOCAL dependent LDS reads while the other wavefronts waited at the
    // barrier; a whole wavefront per rank was measured before and is slower still: the ranks queue up.)
    const u32 count = work_count;
    const u32 sub = threadIdx.x & 7u;
    for (u32 wi = threadIdx.x >> 3; wi < count; wi += BLOCK / 8) {
        const u32 local = work[wi], at = ANN_HALO + local;
        const u32 v = tile[at];
        constexpr u32 NONE_D = ANN_LOCAL + 1;
        u32 d = NONE_D;
#pragma unroll
        for (u32 t = 0; t < (ANN_LOCAL + 7) / 8; t++) {
            const u32 dist = 1u + sub + 8u * t;
            if (dist <= ANN_LOCAL && d == NONE_D && tile[at - dist] <= v) d = dist;
        }
        d = min(d, (u32)__shfl_xor((int)d, 1, 8));
        d = min(d, (u32)__shfl_xor((int)d, 2, 8));
        d = min(d, (u32)__shfl_xor((int)d, 4, 8));
        bool far = d == NONE_D;
        u32 a = 0;
        if (!far && tile[at - d] < v) {                     // first l-index: width = NSV - PSV
            u32 e = NONE_D;
#pragma unroll
            for (u32 t = 0; t < (ANN_LOCAL + 7) / 8; t++) {
                const u32 dist = 1u + sub + 8u * t;
                if (dist <= ANN_LOCAL && e == NONE_D && tile[at + dist] < v) e = dist;
            }
            e = min(e, (u32)__shfl_xor((int)e, 1, 8));
            e = min(e, (u32)__shfl_xor((int)e, 2, 8));
            e = min(e, (u32)__shfl_xor((int)e, 4, 8));
            if (e == NONE_D) far = true;
            else a = d + e;
        }
        if (sub == 0) {
            // (the tile's own stretch of the list: a single counter for all workgroups would serialise them)
            if (far) wide_list[tile_base + atomicAdd(&far_count, 1u)] = tile_base + local;
            else ann[tile_base + local] = a;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) wide_count[blockIdx.x] = far_count;
}

#define ANN_WIDE_SLOTS 16                // threads per tile in ann_wide_kernel
__global__ __launch_bounds__(BLOCK) void ann_wide_kernel(Pyramid P, u32 n, u32 n_tiles, const u32 *__restrict__ wide_list,
                                                         const u32 *__restrict__ wide_count, u32 *__restrict__ ann)
{
    const u32 *lcp = P.ptr[0];
    const u32 tile = blockIdx.x * (BLOCK / ANN_WIDE_SLOTS) + threadIdx.x / ANN_WIDE_SLOTS;
    if (tile >= n_tiles) return;
    const u32 count = wide_count[tile];
    for (u32 i = threadIdx.x % ANN_WIDE_SLOTS; i < count; i += ANN_WIDE_SLOTS) {
        const u32 k = wide_list[tile * ANN_TILE + i];
        const u32 v = lcp[k];
        u32 pse, nsv;                                           // pse exists: the segment start holds 0;
        pyr_find_both(P, k, v, pse, nsv);                       // the search to the right stops at the next segment start
        u32 a = 0;
        // (pse is only missing when the table is not an LCP table: a speculative build that guessed wrong)
        if (pse != NONE_U32 && lcp[pse] < v) a = (nsv == NONE_U32 ? n : nsv) - pse;    // first l-index: pse == PSV, width = NSV - PSV
        ann[k] = a;
    }
}

// The three child tables, positions local to the document, 0 = none (easa.py:268-304), in two launches like the
// annotation table.  child_stream_kernel stages 1024 LCP values + CH_HALO to either side in LDS.  For every rank: the
// nearest value <= its own to the left (PSE) and to the right (NSE) and the leftmost minimum of what lies in between --
// that is up[] and down[] --; next[] is NSE where the values are equal.  Phase 1 decides a thread's 4 ranks from the
// CH_NEAR neighbours to either side out of registers; what lies further away goes to an LDS work list and is looked at
// by eight lanes per rank, at most CH_LOCAL ranks to either side (round 4; before, every thread walked up to 24 ranks
// from each of its 4 ranks, and with a few per cent of wide intervals nearly every wavefront waited for a full-length
// walk at every one of the four: 0.59 ms for the 64 MiB document).  Ranks wider still (the top of the tree) go to the
// tile's list for child_wide_kernel, which uses the pyramid.
#define CH_NEAR 8
#ifndef CH_HALO
#define CH_HALO 64
#endif
#ifndef CH_LOCAL
#define CH_LOCAL 64
#endif
#define CH_IPT 4
#define CH_TILE (BLOCK * CH_IPT)

__device__ __forceinline__ void child_of_rank(const Pyramid &P, u32 k, u32 seg, u32 seg_end, u32 &u, u32 &dn, u32 &nx)
{
    const u32 *lcp = P.ptr[0];
    const u32 v = lcp[k];
    // previous / next position with a value <= v, inside the document
    u32 pse = NONE_U32, nse = NONE_U32;
    if (k > seg) {
        pse = pyr_find_left<false>(P, k, v);
        if (pse != NONE_U32 && pse < seg) pse = NONE_U32;
    }
    nse = pyr_find_right<false>(P, k, v);
    if (nse != NONE_U32 && nse >= seg_end) nse = NONE_U32;
    nx = (nse != NONE_U32 && lcp[nse] == v) ? nse - seg : 0u;
    u = dn = 0;
    if (pse != NONE_U32 && k - pse > 1) u = pyr_leftmost_argmin(P, pse, k) - seg;
    if (nse != NONE_U32 && nse - k > 1) dn = pyr_leftmost_argmin(P, k, nse) - seg;
}

__global__ __launch_bounds__(BLOCK) void child_stream_kernel(Pyramid P, const u32 *__restrict__ doc_off, u32 n_docs, u32 n,
                                                             u32 *__restrict__ up, u32 *__restrict__ down,
                                                             u32 *__restrict__ next, u32 *__restrict__ wide_list,
                                                             u32 *__restrict__ wide_count)
{
    __shared__ __attribute__((aligned(16))) u32 tile[CH_TILE + 2 * CH_HALO];
    __shared__ u32 work[CH_TILE];
    __shared__ u32 far_count, work_count;
    __shared__ u32 tile_doc[3];                         // the document of the tile's first rank, its first rank, its end
    const u32 *lcp = P.ptr[0];
    const u32 tile_base = blockIdx.x * CH_TILE;
    if (threadIdx.x == 0) {
        far_count = 0; work_count = 0;
        // (one binary search per workgroup instead of one per thread: nearly every rank of the tile lies in this document)
        const u32 d = n_docs > 1 ? doc_of(doc_off, n_docs, tile_base < n ? tile_base : n - 1u) : 0u;
        tile_doc[0] = d;
        tile_doc[1] = n_docs > 1 ? doc_off[d] : 0u;
        tile_doc[2] = n_docs > 1 ? doc_off[d + 1] : n;
    }
    const u32 k0 = tile_base + threadIdx.x * CH_IPT;
    const u32 padded = (n + PYR_FAN - 1u) & ~(PYR_FAN - 1u);
    {   // ranks outside [0, n) read as 0, which ends every walk
        uint4 x = {0u, 0u, 0u, 0u};
        if (k0 < padded) x = *reinterpret_cast<const uint4 *>(lcp + k0);
        if (k0 + 0 >= n) x.x = 0u;
        if (k0 + 1 >= n) x.y = 0u;
        if (k0 + 2 >= n) x.z = 0u;
        if (k0 + 3 >= n) x.w = 0u;
        // (the halo is requested before the stretch is staged: one round trip for both)
        const bool has_halo = threadIdx.x < 2 * CH_HALO / 4;
        const bool left = threadIdx.x < CH_HALO / 4;
        const u32 q = left ? threadIdx.x : threadIdx.x - CH_HALO / 4;
        const i64 g = left ? (i64)tile_base - CH_HALO + 4 * q : (i64)tile_base + CH_TILE + 4 * q;
        uint4 h = {0u, 0u, 0u, 0u};
        if (has_halo && g >= 0 && g < (i64)padded) h = *reinterpret_cast<const uint4 *>(lcp + g);
        *reinterpret_cast<uint4 *>(&tile[CH_HALO + threadIdx.x * CH_IPT]) = x;
        if (has_halo) {
            if (g + 0 >= (i64)n) h.x = 0u;
            if (g + 1 >= (i64)n) h.y = 0u;
            if (g + 2 >= (i64)n) h.z = 0u;
            if (g + 3 >= (i64)n) h.w = 0u;
            *reinterpret_cast<uint4 *>(&tile[left ? 4 * q : CH_HALO + CH_TILE + 4 * q]) = h;
        }
    }
    __syncthreads();
    // phase 1: every thread decides its 4 ranks from the CH_NEAR neighbours to either side (registers); a rank whose PSE
    // or NSE lies further away goes to the work list
    if (k0 < n) {
        // the document of the thread's first rank (the others step forward from it)
        u32 d = tile_doc[0], seg = tile_doc[1], seg_end = tile_doc[2];
        while (k0 >= seg_end) { d++; seg = seg_end; seg_end = doc_off[d + 1]; }
        u32 c[CH_IPT + 2 * CH_NEAR];
#pragma unroll
        for (int q = 0; q < (CH_IPT + 2 * CH_NEAR) / 4; q++) {
            const uint4 x = *reinterpret_cast<const uint4 *>(&tile[CH_HALO + threadIdx.x * CH_IPT - CH_NEAR + 4 * q]);
            c[4 * q] = x.x; c[4 * q + 1] = x.y; c[4 * q + 2] = x.z; c[4 * q + 3] = x.w;
        }
        u32 o_up[CH_IPT], o_dn[CH_IPT], o_nx[CH_IPT];
        u32 later = 0;                                      // bit e: rank k0 + e is left to phase 2
#pragma unroll
        for (int e = 0; e < CH_IPT; e++) {
            const u32 k = k0 + e;
            o_up[e] = o_dn[e] = o_nx[e] = 0;
            if (k >= n) break;
            while (k >= seg_end) { d++; seg = seg_end; seg_end = doc_off[d + 1]; }
            const int i = CH_NEAR + e;                      // c[i] = lcp[k]
            const u32 v = c[i];
            // to the left: nearest value <= v (PSE) and the leftmost minimum of the run before it
            if (k > seg) {
                u32 j = 0, best = NONE_U32, bj = 0;
#pragma unroll
                for (int t = 1; t <= CH_NEAR; t++) {
                    const u32 x = c[i - t];
                    if (j == 0) {
                        if (x <= v) j = (u32)t;
                        else if (x <= best) { best = x; bj = (u32)t; }   // (<=: the leftmost of equal minima)
                    }
                }
                if (j == 0) { later |= 1u << e; continue; }
                if (j > 1) o_up[e] = k - bj - seg;          // (pse = k - j >= seg: the document's first rank holds 0)
            }
            // to the right: nearest value <= v (NSE), inside the document, and the leftmost minimum before it
            {
                u32 j = 0, best = NONE_U32, bj = 0;
#pragma unroll
                for (int t = 1; t <= CH_NEAR; t++) {
                    const u32 x = c[i + t];
                    if (j == 0) {
                        if (x <= v) j = (u32)t;
                        else if (x < best) { best = x; bj = (u32)t; }
                    }
                }
                if (j == 0) { later |= 1u << e; continue; }
                if (k + j < seg_end) {                      // (a rank of the next document, or past the end: no NSE)
                    if (c[i + j] == v) o_nx[e] = k + j - seg;
                    if (j > 1) o_dn[e] = k + bj - seg;
                }
            }
        }
        if (later == 0 && k0 + CH_IPT <= n) {
            *reinterpret_cast<uint4 *>(up + k0) = uint4{o_up[0], o_up[1], o_up[2], o_up[3]};
            *reinterpret_cast<uint4 *>(down + k0) = uint4{o_dn[0], o_dn[1], o_dn[2], o_dn[3]};
            *reinterpret_cast<uint4 *>(next + k0) = uint4{o_nx[0], o_nx[1], o_nx[2], o_nx[3]};
        } else {
#pragma unroll
            for (int e = 0; e < CH_IPT; e++) {
                if (k0 + e >= n) break;
                if ((later >> e) & 1u) { work[atomicAdd(&work_count, 1u)] = threadIdx.x * CH_IPT + e; continue; }
                up[k0 + e] = o_up[e];
                down[k0 + e] = o_dn[e];
                next[k0 + e] = o_nx[e];
            }
        }
    }
    __syncthreads();
    // phase 2: eight lanes per rank (as in ann_stream_kernel) -- lane q looks at the distances q + 1, q + 9, ... up to CH_LOCAL;
    // the nearest value <= v comes from three shuffles, the leftmost minimum in front of it from three more (value and
    // distance in one 64-bit word: equal minima are told apart by the distance)
    const u32 count = work_count;
    const u32 sub = threadIdx.x & 7u;
    for (u32 wi = threadIdx.x >> 3; wi < count; wi += BLOCK / 8) {
        const u32 local = work[wi], at = CH_HALO + local, k = tile_base + local;
        const u32 v = tile[at];
        u32 seg = tile_doc[1], seg_end = tile_doc[2];
        if (k >= seg_end) {
            u32 d = tile_doc[0] + 1u;
            while (k >= doc_off[d + 1]) d++;
            seg = doc_off[d];
            seg_end = doc_off[d + 1];
        }
        constexpr u32 NONE_D = CH_LOCAL + 1;
        auto group_min = [](u32 x) -> u32 {
            x = min(x, (u32)__shfl_xor((int)x, 1, 8));
            x = min(x, (u32)__shfl_xor((int)x, 2, 8));
            return min(x, (u32)__shfl_xor((int)x, 4, 8));
        };
        auto group_min64 = [](u64 x) -> u64 {
#pragma unroll
            for (int m = 1; m <= 4; m <<= 1) {
                const u64 y = ((u64)(u32)__shfl_xor((int)(u32)(x >> 32), m, 8) << 32) | (u32)__shfl_xor((int)(u32)x, m, 8);
                x = y < x ? y : x;
            }
            return x;
        };
        bool far = false;
        u32 u = 0, dn = 0, nx = 0;
        if (k > seg) {
            u32 j = NONE_D;
#pragma unroll
            for (u32 t = 0; t < (CH_LOCAL + 7) / 8; t++) {
                const u32 dist = 1u + sub + 8u * t;
                if (dist <= CH_LOCAL && j == NONE_D && tile[at - dist] <= v) j = dist;
            }
            j = group_min(j);
            if (j == NONE_D) far = true;
            else if (j > 1) {
                // the leftmost minimum of the ranks k - j + 1 .. k - 1: the smallest value, of equal ones the LARGEST distance
                u64 best = ~0ull;
#pragma unroll
                for (u32 t = 0; t < (CH_LOCAL + 7) / 8; t++) {
                    const u32 dist = 1u + sub + 8u * t;
                    if (dist < j) {
                        const u64 w = ((u64)tile[at - dist] << 32) | (u32)(~dist);
                        best = w < best ? w : best;
                    }
                }
                best = group_min64(best);
                u = k - ~(u32)best - seg;
            }
        }
        if (!far) {
            u32 j = NONE_D;
#pragma unroll
            for (u32 t = 0; t < (CH_LOCAL + 7) / 8; t++) {
                const u32 dist = 1u + sub + 8u * t;
                if (dist <= CH_LOCAL && j == NONE_D && tile[at + dist] <= v) j = dist;
            }
            j = group_min(j);
            if (j == NONE_D) far = true;
            else if (k + j < seg_end) {                     // (a rank of the next document, or past the end: no NSE)
                if (tile[at + j] == v) nx = k + j - seg;
                if (j > 1) {
                    u64 best = ~0ull;                       // ... of equal minima the SMALLEST distance
#pragma unroll
                    for (u32 t = 0; t < (CH_LOCAL + 7) / 8; t++) {
                        const u32 dist = 1u + sub + 8u * t;
                        if (dist < j) {
                            const u64 w = ((u64)tile[at + dist] << 32) | dist;
                            best = w < best ? w : best;
                        }
                    }
                    best = group_min64(best);
                    dn = k + (u32)best - seg;
                }
            }
        }
        if (sub == 0) {
            if (far) wide_list[tile_base + atomicAdd(&far_count, 1u)] = k;
            else { up[k] = u; down[k] = dn; next[k] = nx; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) wide_count[blockIdx.x] = far_count;
}

#define CH_WIDE_SLOTS 16
__global__ __launch_bounds__(BLOCK) void child_wide_kernel(Pyramid P, const u32 *__restrict__ doc_off, u32 n_docs, u32 n,
                                                           u32 n_tiles, const u32 *__restrict__ wide_list,
                                                           const u32 *__restrict__ wide_count, u32 *__restrict__ up,
                                                           u32 *__restrict__ down, u32 *__restrict__ next)
{
    const u32 tile = blockIdx.x * (BLOCK / CH_WIDE_SLOTS) + threadIdx.x / CH_WIDE_SLOTS;
    if (tile >= n_tiles) return;
    const u32 count = wide_count[tile];
    for (u32 i = threadIdx.x % CH_WIDE_SLOTS; i < count; i += CH_WIDE_SLOTS) {
        const u32 k = wide_list[tile * CH_TILE + i];
        u32 seg = 0, seg_end = n;
        if (n_docs > 1) {
            const u32 d = doc_of(doc_off, n_docs, k);
            seg = doc_off[d];
            seg_end = doc_off[d + 1];
        }
        u32 u, dn, nx;
        child_of_rank(P, k, seg, seg_end, u, dn, nx);
        up[k] = u;
        down[k] = dn;
        next[k] = nx;
    }
}

// Left boundaries of the lcp-intervals (easa.py:38-85, the traversals): an lcp-interval l-[i..j], l > 0, is
// named by its first l-index k (the rank with anntab[k] > 0): i = PSV(k), j = i + anntab[k] - 1, l = lcptab[k]
// (SURVEY.md Appendix A.2).  left[k] = PSV(k) local to the document for those ranks, NONE_U32 for every other.
// The host turns the intervals into the reference's pre-/post-order visits (east/asts/easa_hip.py).
__global__ __launch_bounds__(BLOCK) void interval_left_kernel(Pyramid P, const u32 *__restrict__ ann, u32 seg, u32 nd,
                                                              u32 *__restrict__ left)
{
    const u32 r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= nd) return;
    const u32 k = seg + r;
    u32 out = NONE_U32;
    if (r > 0 && ann[k] > 0) {
        const u32 v = P.ptr[0][k];                          // > 0: the document's first rank (lcp 0) bounds the search
        out = pyr_find_left<true>(P, k, v) - seg;
    }
    left[r] = out;
}
