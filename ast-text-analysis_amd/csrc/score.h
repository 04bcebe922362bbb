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

// raw query code points -> dense codes of the corpus alphabet
__global__ __launch_bounds__(BLOCK) void query_map_kernel(const u32 *__restrict__ q_raw, u32 n_q,
                                                          const u32 *__restrict__ code_map,
                                                          u32 *__restrict__ q_code)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_q) return;
    const u32 c = q_raw[i];
    u32 code = Q_NOMATCH;
    if (c < EAST_HIP_TERMINATOR_START) {
        code = code_map[c];
        if (code == 0) code = Q_NOMATCH;          // symbol absent from the corpus
    }
    q_code[i] = code;
}

__global__ __launch_bounds__(BLOCK) void score_walk_kernel(
    const u32 *__restrict__ s, const u32 *__restrict__ sa, const u32 *__restrict__ doc_off,
    const u32 *__restrict__ n_strings, u32 n_docs, const u32 *__restrict__ q_code,
    const u32 *__restrict__ q_end, u32 n_q, int normalized, double *__restrict__ suffix_out)
{
    const u64 gid = (u64)blockIdx.x * BLOCK + threadIdx.x;
    if (gid >= (u64)n_docs * n_q) return;
    const u32 d = (u32)(gid / n_q);
    const u32 si = (u32)(gid - (u64)d * n_q);
    const u32 seg = doc_off[d];
    const u32 nd = doc_off[d + 1] - seg;
    const u32 root_ann = nd - n_strings[d];
    const u32 *sad = sa + seg;
    const u32 end = q_end[si];

    u32 lo = 0, hi = nd - 1, depth = 0, nodes = 0;
    double acc = 0.0;
    for (u32 t = si; t < end; t++) {
        const u32 c = q_code[t];
        if (c == Q_NOMATCH) break;
        u32 a, b;
        if (lo == hi) {
            if (s[sad[lo] + depth] != c) break;
            a = b = lo;
        } else {
            u32 x = lo, y = hi + 1;                       // lower bound of c at this depth
            while (x < y) {
                const u32 mid = (x + y) >> 1;
                if (s[sad[mid] + depth] < c) x = mid + 1; else y = mid;
            }
            a = x;
            y = hi + 1;                                   // upper bound, from a
            while (x < y) {
                const u32 mid = (x + y) >> 1;
                if (s[sad[mid] + depth] <= c) x = mid + 1; else y = mid;
            }
            if (x == a) break;                            // no suffix continues with c
            b = x - 1;
        }
        if (b - a < hi - lo) {                            // a child node was entered
            const u32 parent = depth == 0 ? root_ann : hi - lo + 1;
            acc += (double)(b - a + 1) / (double)parent;
            nodes++;
        }
        lo = a; hi = b; depth++;
    }
    double r = 0.0;
    if (depth > 0) {
        r = (acc + (double)depth) - (double)nodes;        // easa.py:127
        if (normalized) r /= (double)depth;               // easa.py:128-129
    }
    suffix_out[(u64)d * n_q + si] = r;
}

// out[k*D + d] = (sum of the keyphrase's suffix results, in suffix order) / |q|
__global__ __launch_bounds__(BLOCK) void score_reduce_kernel(const double *__restrict__ suffix,
                                                             const u32 *__restrict__ q_off,
                                                             u32 n_kp, u32 n_docs, u32 n_q,
                                                             double *__restrict__ out)
{
    const u64 gid = (u64)blockIdx.x * BLOCK + threadIdx.x;
    if (gid >= (u64)n_kp * n_docs) return;
    const u32 k = (u32)(gid / n_docs);
    const u32 d = (u32)(gid - (u64)k * n_docs);
    const u32 b = q_off[k], e = q_off[k + 1];
    const double *row = suffix + (u64)d * n_q;
    double total = 0.0;
    for (u32 i = b; i < e; i++) total += row[i];          // easa.py:130
    out[gid] = total / (double)(e - b);                   // easa.py:134
}
