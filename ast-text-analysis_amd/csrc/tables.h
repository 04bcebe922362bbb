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
// Nearest-smaller-value and range-minimum queries run on a 64-ary min pyramid
// over lcptab (level i+1 = min of 64 entries of level i, one wavefront
// reduction per entry), so every query is O(64 * log64 n) worst case and a
// handful of L2-resident loads in the common case; the sequential stacks of
// the reference disappear.
#pragma once
#include "common.h"

#define PYR_MAX_LEVELS 7
#define NONE_U32 0xFFFFFFFFu

struct Pyramid {
    const u32 *ptr[PYR_MAX_LEVELS];
    u32 len[PYR_MAX_LEVELS];
    int levels;
};

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

__global__ __launch_bounds__(BLOCK) void doc_keys_kernel(const u32 *__restrict__ sa,
                                                         const u32 *__restrict__ doc_off,
                                                         u32 n_docs, u32 n, u32 *__restrict__ keys)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) keys[i] = doc_of(doc_off, n_docs, sa[i]);
}

// seg_start[r] = 1 iff rank r is the first rank of a document segment
__global__ __launch_bounds__(BLOCK) void lcp_kernel(const u32 *__restrict__ s,
                                                    const u32 *__restrict__ sa,
                                                    const u32 *__restrict__ doc_off, u32 n_docs,
                                                    u32 n, u32 *__restrict__ lcp)
{
    const u32 r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    if (r == 0) { lcp[0] = 0; return; }
    const u32 i = sa[r - 1], j = sa[r];
    u32 h = 0;
    // different documents never share a terminator, same document: both
    // suffixes end in distinct terminators => the comparison always stops in
    // bounds.  Four symbols per step (8 independent loads in flight); the three
    // pad words behind the stream make the look-ahead safe.
    while (true) {
        const u32 a0 = s[i + h], a1 = s[i + h + 1], a2 = s[i + h + 2], a3 = s[i + h + 3];
        const u32 b0 = s[j + h], b1 = s[j + h + 1], b2 = s[j + h + 2], b3 = s[j + h + 3];
        if (a0 != b0) break;
        if (a1 != b1) { h += 1; break; }
        if (a2 != b2) { h += 2; break; }
        if (a3 != b3) { h += 3; break; }
        h += 4;
    }
    // the first rank of a document compares against the previous document's
    // last suffix; the reference table starts every document with 0
    if (n_docs > 1 && h > 0) {
        const u32 d = doc_of(doc_off, n_docs, r);
        if (doc_off[d] == r) h = 0;
    }
    lcp[r] = h;
}

__global__ __launch_bounds__(BLOCK) void pyramid_level_kernel(const u32 *__restrict__ in, u32 len_in,
                                                              u32 *__restrict__ out)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    const u32 v = wave_min(i < len_in ? in[i] : NONE_U32);
    if (lane_id() == 0 && i < len_in) out[i >> 6] = v;
}

// largest p < k with level0[p] < v (STRICT) or <= v; NONE_U32 if none
template <bool STRICT>
__device__ __forceinline__ u32 pyr_find_left(const Pyramid &P, u32 k, u32 v)
{
    u32 pos = k;
    int lvl = 0;
    while (true) {
        const u32 *M = P.ptr[lvl];
        const u32 start = pos & ~63u;
        u32 q = pos;
        bool found = false;
        while (q > start) {
            q--;
            const u32 x = M[q];
            if (STRICT ? x < v : x <= v) { found = true; break; }
        }
        if (found) {
            while (lvl > 0) {
                lvl--;
                const u32 *C = P.ptr[lvl];
                const u32 base = q << 6;
                u32 c = base + 64u < P.len[lvl] ? base + 64u : P.len[lvl];
                while (c > base) {
                    c--;
                    const u32 x = C[c];
                    if (STRICT ? x < v : x <= v) break;
                }
                q = c;
            }
            return q;
        }
        if (start == 0 || lvl + 1 >= P.levels) return NONE_U32;
        pos = start >> 6;
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
        const u32 *M = P.ptr[lvl];
        const u32 len = P.len[lvl];
        u32 end = (pos + 63u) & ~63u;
        if (end > len) end = len;
        u32 q = pos;
        bool found = false;
        while (q < end) {
            const u32 x = M[q];
            if (STRICT ? x < v : x <= v) { found = true; break; }
            q++;
        }
        if (found) {
            while (lvl > 0) {
                lvl--;
                const u32 *C = P.ptr[lvl];
                const u32 base = q << 6;
                const u32 lim = base + 64u < P.len[lvl] ? base + 64u : P.len[lvl];
                u32 c = base;
                while (c < lim) {
                    const u32 x = C[c];
                    if (STRICT ? x < v : x <= v) break;
                    c++;
                }
                q = c;
            }
            return q;
        }
        if (end >= len || lvl + 1 >= P.levels) return NONE_U32;
        pos = end >> 6;
        lvl++;
    }
}

// leftmost position of the minimum of level0 over the open interval (a, b),
// a + 1 < b.  Walks whole 64-groups through the pyramid.
__device__ __forceinline__ u32 pyr_leftmost_argmin(const Pyramid &P, u32 a, u32 b)
{
    // 1. minimum value over (a, b)
    u32 lo = a + 1, hi = b;           // [lo, hi)
    u32 best = NONE_U32;
    {
        u32 l = lo, h = hi;
        int lvl = 0;
        while (l < h) {
            const u32 *M = P.ptr[lvl];
            // peel unaligned heads/tails at this level, then go up
            while (l < h && (l & 63u)) { const u32 x = M[l]; best = x < best ? x : best; l++; }
            while (l < h && (h & 63u)) { h--; const u32 x = M[h]; best = x < best ? x : best; }
            if (l >= h) break;
            if (lvl + 1 >= P.levels) {
                for (u32 q = l; q < h; q++) { const u32 x = M[q]; best = x < best ? x : best; }
                break;
            }
            l >>= 6; h >>= 6; lvl++;
        }
    }
    // 2. first position >= lo holding a value <= best (it is == best and < hi)
    if (P.ptr[0][lo] == best) return lo;
    return pyr_find_right<false>(P, lo, best);
}

// anntab of one rank (easa.py:306-331).  doc segment starts carry lcp == 0,
// which bounds every search inside the document.
__global__ __launch_bounds__(BLOCK) void ann_kernel(Pyramid P, const u32 *__restrict__ doc_off,
                                                    const u32 *__restrict__ n_strings, u32 n_docs, u32 n,
                                                    u32 *__restrict__ ann)
{
    const u32 k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n) return;
    const u32 *lcp = P.ptr[0];
    const u32 v = lcp[k];
    u32 a = 0;
    if (v == 0) {
        // only a document's first rank carries an annotation here: the root, n_d - m_d
        const u32 d = n_docs > 1 ? doc_of(doc_off, n_docs, k) : 0u;
        if (doc_off[d] == k) a = (doc_off[d + 1] - k) - n_strings[d];
    } else {
        // first l-index of its interval <=> the previous value <= v is strictly smaller
        const u32 pse = pyr_find_left<false>(P, k, v);          // exists: the segment start holds 0
        if (lcp[pse] < v) {
            u32 nsv = pyr_find_right<true>(P, k, v);            // stops at the next segment start
            if (nsv == NONE_U32) nsv = n;
            a = nsv - pse;                                      // pse == PSV here
        }
    }
    ann[k] = a;
}

// The three child tables of one rank, positions local to the document, 0 = none
// (easa.py:268-304).  Not needed by the score walk: computed on first request.
__global__ __launch_bounds__(BLOCK) void child_kernel(Pyramid P, const u32 *__restrict__ doc_off, u32 n_docs,
                                                      u32 n, u32 *__restrict__ up, u32 *__restrict__ down,
                                                      u32 *__restrict__ next)
{
    const u32 k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n) return;
    const u32 *lcp = P.ptr[0];
    const u32 v = lcp[k];
    u32 seg = 0, seg_end = n;
    if (n_docs > 1) {
        const u32 d = doc_of(doc_off, n_docs, k);
        seg = doc_off[d];
        seg_end = doc_off[d + 1];
    }
    // previous / next position with a value <= v, inside the document
    u32 pse = NONE_U32, nse = NONE_U32;
    if (k > seg) {
        pse = pyr_find_left<false>(P, k, v);
        if (pse != NONE_U32 && pse < seg) pse = NONE_U32;
    }
    nse = pyr_find_right<false>(P, k, v);
    if (nse != NONE_U32 && nse >= seg_end) nse = NONE_U32;
    next[k] = (nse != NONE_U32 && lcp[nse] == v) ? nse - seg : 0u;
    u32 u = 0, dn = 0;
    if (pse != NONE_U32 && k - pse > 1) u = pyr_leftmost_argmin(P, pse, k) - seg;
    if (nse != NONE_U32 && nse - k > 1) dn = pyr_leftmost_argmin(P, k, nse) - seg;
    up[k] = u;
    down[k] = dn;
}
