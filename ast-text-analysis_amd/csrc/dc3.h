// dc3.h -- data-parallel DC3 / skew suffix array construction (Karkkainen &
// Sanders 2003), the algorithm the reference runs in _kark_sort
// (east/asts/easa.py:155-228), re-designed for the GPU:
//
//   1. sample positions i mod 3 != 0 become packed (s[i],s[i+1],s[i+2]) keys,
//      read coalesced from the symbol stream (no gathers), and are sorted by the
//      LDS-staged wave64 radix sort (radix_sort.h)            [easa.py:163-167]
//   2. naming = inclusive scan of "key differs from predecessor"  [easa.py:169-182]
//   3. if names are not unique: recurse on the name string, all on device,
//      the host only reads one word per level                  [easa.py:184-190]
//   4. non-sample suffixes: stable compaction of the mod-1 entries of SA12
//      (scan) + one radix sort by first symbol                 [easa.py:192-194]
//   5. merge-path merge of the two sorted sets with the DC3 comparator
//                                                               [easa.py:196-228]
//
// The suffix array is unique for a given symbol order, so the result is
// bit-identical to the reference's suftab although none of its sequential
// dict-counting / list-merging code is reproduced.
//
// Symbol string convention: s[0..n) in [1, sigma], s[n..n+3) == 0.
#pragma once
#include "common.h"
#include "radix_sort.h"
#include "scan.h"

__device__ __forceinline__ u32 dc3_sample_pos(u32 t, u32 n0)
{
    return t < n0 ? 3u * t + 1u : 3u * (t - n0) + 2u;
}

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

// ---- step 3: ranks from the recursive suffix array --------------------------
__global__ __launch_bounds__(BLOCK) void dc3_rank_kernel(const u32 *__restrict__ sa12, u32 n02,
                                                         u32 *__restrict__ s12)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n02) s12[sa12[i]] = i + 1;
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

// ---- step 5: merge -----------------------------------------------------------
// true iff the sample suffix (index t in s12 space) sorts before the non-sample
// suffix at text position j (j mod 3 == 0).            [easa.py:202-207]
__device__ __forceinline__ bool dc3_sample_leq(const u32 *__restrict__ s,
                                               const u32 *__restrict__ s12, u32 n0, u32 t, u32 j)
{
    if (t < n0) {
        const u32 i = 3u * t + 1u;
        const u32 a = s[i], c = s[j];
        if (a != c) return a < c;
        return s12[t + n0] <= s12[j / 3u];
    }
    const u32 i = 3u * (t - n0) + 2u;
    u32 a = s[i], c = s[j];
    if (a != c) return a < c;
    a = s[i + 1];
    c = s[j + 1];
    if (a != c) return a < c;
    return s12[t - n0 + 1u] <= s12[j / 3u + n0];
}

#define MERGE_IPT 8     // outputs per thread

__global__ __launch_bounds__(BLOCK) void dc3_merge_kernel(const u32 *__restrict__ s,
                                                          const u32 *__restrict__ s12,
                                                          const u32 *__restrict__ sa12,   // n02
                                                          const u32 *__restrict__ sa0,    // n0
                                                          u32 n, u32 n0, u32 n02, u32 skip,
                                                          u32 *__restrict__ sa_out)
{
    const u32 k0 = (blockIdx.x * BLOCK + threadIdx.x) * MERGE_IPT;
    if (k0 >= n) return;
    const u32 *A = sa12 + skip;
    const u32 nA = n02 - skip, nB = n0;
    // merge-path split of diagonal k0
    u32 lo = k0 > nB ? k0 - nB : 0u;
    u32 hi = k0 < nA ? k0 : nA;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (dc3_sample_leq(s, s12, n0, A[mid], sa0[k0 - 1u - mid])) lo = mid + 1u;
        else hi = mid;
    }
    u32 a = lo, bq = k0 - lo;
#pragma unroll 1
    for (u32 i = 0; i < MERGE_IPT && k0 + i < n; i++) {
        bool take_a;
        if (bq >= nB) take_a = true;
        else if (a >= nA) take_a = false;
        else take_a = dc3_sample_leq(s, s12, n0, A[a], sa0[bq]);
        if (take_a) { sa_out[k0 + i] = dc3_sample_pos(A[a], n0); a++; }
        else { sa_out[k0 + i] = sa0[bq]; bq++; }
    }
}

// ---- host driver ----------------------------------------------------------------
// s: n+3 symbols (three zero pads), values in [1, sigma].  sa_out: n words.
// Returns the number of levels executed.
static int dc3_suffix_array(Ctx &ctx, const u32 *s, u32 n, u32 sigma, u32 *sa_out, int depth = 0)
{
    const u32 n0 = (n + 2) / 3, n1 = (n + 1) / 3, n2 = n / 3, n02 = n0 + n2;
    const int b = bit_width_u32(sigma);
    const u32 g02 = ceil_div_u32(n02, BLOCK);
    int levels = 1;
    Arena &ar = *ctx.arena;
    const size_t mark_level = ar.mark();
    u32 *s12 = ar.alloc<u32>((size_t)n02 + 3);
    u32 *sa12 = ar.alloc<u32>(n02);

    // -- sort the sample triples, name them --------------------------------
    u32 n_names = 0;
    {
        const size_t mark = ar.mark();
        u32 *names = ar.alloc<u32>(n02);
        const u32 *sorted_vals = nullptr;
        if (3 * b <= 32) {
            SortBufs<u32> sb;
            for (int k = 0; k < 2; k++) { sb.keys[k] = ar.alloc<u32>(n02); sb.vals[k] = ar.alloc<u32>(n02); }
            LAUNCH(ctx, (dc3_triple_keys_kernel<u32>), g02, s, n0, n02, b, sb.keys[0], sb.vals[0]);
            const int r = radix_sort_pairs<u32>(ctx, sb, n02, 3 * b);
            device_scan<KeyNeqIn<u32>, true>(ctx, KeyNeqIn<u32>{sb.keys[r]}, n02, names);
            sorted_vals = sb.vals[r];
        } else if (3 * b <= 64) {
            SortBufs<u64> sb;
            for (int k = 0; k < 2; k++) { sb.keys[k] = ar.alloc<u64>(n02); sb.vals[k] = ar.alloc<u32>(n02); }
            LAUNCH(ctx, (dc3_triple_keys_kernel<u64>), g02, s, n0, n02, b, sb.keys[0], sb.vals[0]);
            const int r = radix_sort_pairs<u64>(ctx, sb, n02, 3 * b);
            device_scan<KeyNeqIn<u64>, true>(ctx, KeyNeqIn<u64>{sb.keys[r]}, n02, names);
            sorted_vals = sb.vals[r];
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
        if (n_names == n02 && !ctx.dry)   // unique names: the sorted order is SA12 already
            HIP_CHECK(hipMemcpyAsync(sa12, sorted_vals, (size_t)n02 * sizeof(u32),
                                     hipMemcpyDeviceToDevice, ctx.stream));
        ar.release(mark);
    }

    // -- recurse on the name string ----------------------------------------
    if (n_names < n02) {
        levels += dc3_suffix_array(ctx, s12, n02, n_names, sa12, depth + 1);
        LAUNCH(ctx, dc3_rank_kernel, g02, (const u32 *)sa12, n02, s12);
    }

    // -- non-sample suffixes + merge ------------------------------------------
    {
        u32 *slot = ar.alloc<u32>(n02);
        device_scan<LtIn, false>(ctx, LtIn{sa12, n0}, n02, slot);
        SortBufs<u32> s0;
        for (int k = 0; k < 2; k++) { s0.keys[k] = ar.alloc<u32>(n0); s0.vals[k] = ar.alloc<u32>(n0); }
        LAUNCH(ctx, dc3_compact_s0_kernel, g02, s, (const u32 *)sa12, (const u32 *)slot, n0, n02,
               s0.keys[0], s0.vals[0]);
        const int r0 = radix_sort_pairs<u32>(ctx, s0, n0, b);
        LAUNCH(ctx, dc3_merge_kernel, ceil_div_u32(ceil_div_u32(n, MERGE_IPT), BLOCK), s,
               (const u32 *)s12, (const u32 *)sa12, (const u32 *)s0.vals[r0], n, n0, n02, n0 - n1,
               sa_out);
    }
    ar.release(mark_level);
    return levels;
}
