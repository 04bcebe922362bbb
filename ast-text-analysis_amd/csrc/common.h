// common.h -- shared host/device helpers for the gfx950 EASA backend.
//
// Everything here is written for CDNA4 only: 64-lane wavefronts, 256-thread
// workgroups by default (4 waves = one per SIMD; the radix scatter runs 1024-thread
// workgroups to keep 32 waves per CU resident), 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;

#include "../../include/east_hip.h"

// Direct LCP comparison is capped: ranks whose common prefix reaches LCP_DIRECT_CAP symbols are
// marked (LCP_PARTIAL_BIT) and finished by the Kasai-style pass in tables.h, so that a highly
// repetitive input costs O(n + work of the marked ranks) instead of the sum of all LCPs.
#define LCP_DIRECT_CAP (1u << 14)
// ... and a build only lets a BOUNDED number of comparisons go that far: one that reaches LCP_SOFT_CAP symbols asks the
// build's budget (LcpBudget: n / 256 deep comparisons, counted in 64 slots) and stops there once it is spent -- an input
// in which every suffix has a long common prefix with its neighbour (the reference's own benchmark input, analysis/
// utils.py:5-9: 100 identical strings) would otherwise cost n x 16 384 symbol comparisons before the Kasai pass gets its
// turn (measured: 109 of 127 ms for 10 M symbols).  An unfinished entry is LCP_PARTIAL_BIT | the length known to agree.
#define LCP_SOFT_CAP 256u
#define LCP_PARTIAL_BIT 0x80000000u
#define LCP_BUDGET_SLOTS 64u
struct LcpBudget {              // (how it is counted: lcp_deep_allowed below)
    u32 *slots = nullptr;
    u32 per_slot = 0;
};

#define WAVE 64
#define BLOCK 256              // threads per workgroup everywhere
#define WAVES_PER_BLOCK 4

// ---------------------------------------------------------------- errors ----
struct EastError {
    int code;
    std::string msg;
};

[[noreturn]] inline void east_throw(int code, const std::string &msg) { throw EastError{code, msg}; }

#define HIP_CHECK(expr)                                                               \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            char _b[512];                                                             \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr,                  \
                     hipGetErrorString(_e), __FILE__, __LINE__);                      \
            east_throw(_e == hipErrorOutOfMemory ? EAST_HIP_ERR_OOM : EAST_HIP_ERR_HIP, _b); \
        }                                                                             \
    } while (0)

// ----------------------------------------------------------------- arena ----
// Bump allocator over one device allocation with stack discipline
// (mark/release).  In dry mode nothing is backed by memory: the same host
// orchestration code is run to measure the high-water mark before the real
// arena is allocated, so a build never calls hipMalloc in its timed region.
struct Arena {
    char *base = nullptr;
    size_t cap = 0, off = 0, high = 0;
    bool dry = false;

    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
    void *alloc_bytes(size_t bytes)
    {
        size_t a = (off + 255) & ~(size_t)255;
        size_t end = a + bytes;
        if (!dry && end > cap) east_throw(EAST_HIP_ERR_INTERNAL, "device arena exhausted");
        off = end;
        if (off > high) high = off;
        return dry ? (void *)(uintptr_t)(0x1000 + a) : (void *)(base + a);
    }
    template <class T> T *alloc(size_t count) { return (T *)alloc_bytes(count * sizeof(T)); }
};

struct Stats {
    i64 levels = 0, levels_resolved = 0, refine_rounds = 0, window_sorted = 0, merge_elems = 0, radix_passes = 0, radix_elems = 0, radix_elem_bytes = 0;
    i64 radix_elems_u32 = 0, radix_elems_u64 = 0, radix_passes_u32 = 0, radix_passes_u64 = 0;
    i64 long_repeats = 0;       // the placement pass had to be repeated as mark + commit (duplicated passages)
    i64 lds_sorted = 0;         // elements the refinement rounds ordered inside a workgroup's LDS (lds_group_sort.h)
    i64 first_kept = 0, first_n = 0;    // all-suffix window sort: suffixes the placement pass left in large groups, of how many
    i64 fused_finish = 0;       // the last radix digit and the placement ran as one pass in LDS (lvl0_finish_kernel)
    i64 ht_keys = 0;            // the first-level keys held variable-length code words (ht_code.h)
    i64 seg_sort = 0;           // the first-level sort was segmented by document (radix_sort.h: RsSeg)
    i64 persist_rounds = 0;     // rounds run inside ONE launch by the workgroups that keep their tiles (persist_rounds.h)
};

// Optional per-kernel timing with HIP events on the handle's own stream (the
// stream the kernels are launched on); off unless east_hip_profile_enable().
struct Profiler {
    struct Rec { const char *name; hipEvent_t e0, e1; };
    struct Sum { std::string name; i64 count; double ms; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    std::vector<Sum> sums;
    bool enabled = false;
    std::string only;           // not empty: only launches of this kernel are bracketed (two events per launch cost time)
    bool skipped = false;

    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) east_throw(EAST_HIP_ERR_HIP, "hipEventCreate failed");
        return e;
    }
    void begin(const char *name, hipStream_t st)
    {
        skipped = !only.empty() && only != name;
        if (skipped) return;
        Rec r{name, get(), get()};
        (void)hipEventRecord(r.e0, st);
        recs.push_back(r);
    }
    void end(hipStream_t st)
    {
        if (!skipped) (void)hipEventRecord(recs.back().e1, st);
    }
    // call after the stream has been synchronised
    void collect()
    {
        for (auto &r : recs) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, r.e0, r.e1);
            bool found = false;
            for (auto &s : sums)
                if (s.name == r.name) { s.count++; s.ms += ms; found = true; break; }
            if (!found) sums.push_back(Sum{r.name, 1, ms});
            pool.push_back(r.e0);
            pool.push_back(r.e1);
        }
        recs.clear();
    }
    void reset() { collect(); sums.clear(); }
    ~Profiler()
    {
        for (auto &r : recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

// The test knobs (east_hip_debug_*; include/east_hip.h says what each value means).  g_knobs holds the process-wide
// DEFAULTS the setters change; every entry point copies them ONCE when it starts (knobs_snapshot -> Ctx::knobs or a local),
// so a call runs from its first launch to its last on one consistent set of values whatever another thread sets meanwhile
// -- distinct handles stay independent of each other in that respect too (the in-process device groups drive several).
#define SCORE_SCRATCH_BYTES ((size_t)1 << 30)
#define SCORE_GRID_BLOCKS ((u64)1 << 22)
#define TP_RING_SLOT ((size_t)8 << 20)
static int env_int(const char *name, int absent) { const char *v = getenv(name); return v ? atoi(v) : absent; }
struct Knobs {
    bool window_sort = true;                                            // east_hip_debug_set_window_sort
    bool force_lean = false;                                            // ... (2)
    bool force_wide_keys = getenv("EAST_HIP_WIDE_KEYS") != nullptr;     // ... (3 / 5)
    bool fused_finish = getenv("EAST_HIP_NO_FUSED_FINISH") == nullptr;  // ... (4 / 5 / 9 switch it off)
    bool force_fused = getenv("EAST_HIP_FORCE_FUSED") != nullptr;       // ... (6): the fused finish whatever the plan says
    // variable-length first-level keys (ht_code.h): -1 = where the text's symbol statistics promise a symbol more per
    // key, 0 = never, 1 = whenever a code exists (... (7 / 9 / 8); EAST_HIP_HT: A/B timing)
    int ht_mode = env_int("EAST_HIP_HT", -1);
    int seg_mode = env_int("EAST_HIP_SEG", -1);                         // east_hip_debug_set_segmented_sort: -1 by size, 0 never, 1 wherever it can be done
    bool lds_rounds = getenv("EAST_HIP_NO_LDS_ROUNDS") == nullptr;      // east_hip_debug_set_lds_rounds
    bool fused_classify = getenv("EAST_HIP_NO_FUSED_CLASSIFY") == nullptr;      // ... (2): the stand-alone classification pass
    bool persist = getenv("EAST_HIP_NO_PERSIST") == nullptr;            // ... (1 only): small domains finish in one launch (persist_rounds.h)
    bool persist_large = getenv("EAST_HIP_NO_PERSIST_LARGE") == nullptr;    // ... and large ones, several tiles per workgroup (A/B timing)
    bool persist_force_large = getenv("EAST_HIP_PERSIST_FORCE_LARGE") != nullptr;   // east_hip_debug_set_persist: the large form on any domain
    int persist_max_wgs = env_int("EAST_HIP_PERSIST_WGS", 0);          // ... with at most this many workgroups (0: what the device holds)
    size_t rank_bucket_bytes = (size_t)192 << 20;                       // east_hip_debug_set_rank_bucket_bytes
    bool speculate = getenv("EAST_HIP_NO_SPECULATION") == nullptr;      // east_hip_debug_set_speculation
    bool kg_pairs = getenv("EAST_HIP_NO_KG_PAIRS") == nullptr;          // east_hip_debug_set_score_path
    bool kg_pairs_forced = false;                                       // ... (4)
    bool score_fused = getenv("EAST_HIP_SCORE_UNFUSED") == nullptr;     // ... (1 / 3 / 4)
    int score_endgame = env_int("EAST_HIP_SCORE_ENDGAME", 1);           // KgTables::endgame (A/B timing; east_hip_debug_set_score_path 5 switches it off)
    size_t score_scratch_bytes = SCORE_SCRATCH_BYTES;                   // east_hip_debug_set_score_scratch
    u64 score_grid_blocks = SCORE_GRID_BLOCKS;                          // east_hip_debug_set_score_grid
    i64 tp_stream = getenv("EAST_HIP_TEXT_STREAM") ? atoll(getenv("EAST_HIP_TEXT_STREAM")) : -1;   // east_hip_debug_set_text_stream
    int tp_ring = env_int("EAST_HIP_TEXT_RING", -1);                    // east_hip_debug_set_text_ring
    size_t tp_ring_slot = getenv("EAST_HIP_RING_SLOT") ? std::min<size_t>(TP_RING_SLOT, (size_t)std::max(64, atoi(getenv("EAST_HIP_RING_SLOT")))) : TP_RING_SLOT;
    u32 plan_epoch = 1;                                                 // bumped by the knobs that change what a build allocates
};
static Knobs g_knobs;
static std::mutex g_knobs_mutex;
static inline Knobs knobs_snapshot()
{
    std::lock_guard<std::mutex> lock(g_knobs_mutex);
    return g_knobs;
}
template <class F> static inline void knobs_update(F f)
{
    std::lock_guard<std::mutex> lock(g_knobs_mutex);
    f(g_knobs);
}

struct Ctx {
    Knobs knobs = knobs_snapshot();         // the test knobs as they stood when the call began
    hipStream_t stream = nullptr;
    Arena *arena = nullptr;
    bool dry = false;
    bool lean = false;          // no tie-refinement rounds (their buffers did not fit the device)
    // Speculative build (east_hip.hip, build_common): the host does not wait for the device where the previous
    // build on the handle tells it what to expect -- the size of the text alphabet, "no suffix is left in a
    // large tie group after the placement pass".  The device checks both and the one read-back at the
    // end of the build finds out; a wrong guess costs a second, non-speculative build.
    bool spec = false;          // the build does not wait for the alphabet (it uses the last build's)
    bool spec_rounds = false;   // ... nor for the placement pass's counts (it goes on as if no tie group were large)
    u32 *spec_out = nullptr;    // [0] suffixes left in large groups, [1] placement gave up on a long repeat
    u32 *zeroed_word = nullptr; // one word the build has already zeroed: the first level-0 pass takes it for its fail flag
    u32 *kg_bad = nullptr;      // raised by the fused finish when the k-gram marks it writes are incomplete (a bucket handed to the rounds)
    // What the sample of the build's own text says (sample_prefix_kernel; a handle's first build and every build that
    // waits for the alphabet): of sample_n consecutive suffixes, sample_dup2[l] / sample_dup4[l] share their first
    // l symbols with at least 1 / 3 others of the sample.  sample_n == 0: no sample (speculative build: the plan of
    // the build before is taken instead -- plan_wide / plan_fused).
    u32 sample_n = 0, sample_dup2[9] = {0}, sample_dup4[9] = {0};
    u32 rep_n = 0, rep_dup = 0;             // "is the text repetitive?": of rep_n sample suffixes, rep_dup share 8 symbols with three others (also from samples too small to plan from)
    int plan_wide = -1, plan_fused = -1;    // -1 = decide from the sample / the estimates; 0 / 1 = as the build before did
    int did_wide = 0, did_fused = 0;        // out: what the all-suffix window sort did (the next speculative build's plan)
    // An order-preserving variable-length code for the text's symbols is at hand (ht_code.h; ht_max_len > 0): the device
    // tables, the longest code word and the mean code word length over the text (bits per symbol)
    const u32 *ht_enc = nullptr;
    const uint16_t *ht_dec = nullptr;
    int ht_max_len = 0;
    double ht_mean_len = 0.0;
    int plan_ht = -1, did_ht = 0;           // first-level keys of variable-length code words: plan (as plan_wide) / what was done
    int plan_persist = -1, did_persist = 0; // the first domain went to the persistent rounds at once (repetitive text: persist_rounds.h)
    int did_seg = 0;                        // the first-level sort kept the documents apart by segments, not by key bits
    std::vector<u32> seg_host;              // ... its tables on the host (the uploads are asynchronous)
    LcpBudget lcp_budget;                   // the build's budget of deep LCP comparisons (LCP_SOFT_CAP)
    Stats *stats = nullptr;
    Profiler *prof = nullptr;
};

// Wait for a stream the way the build's read-backs want it: they sit between kernels a few microseconds long, and a
// blocking hipStreamSynchronize comes back tens of microseconds after the stream has drained (the waiter sleeps on an
// interrupt).  The stream is polled for a short while first; what is still running after that is waited for asleep.
static bool g_spin_sync = getenv("EAST_HIP_NO_SPIN_SYNC") == nullptr;
static inline hipError_t sync_stream(hipStream_t stream)
{
    if (g_spin_sync) {
        bool waited = false;
        for (int i = 0; i < 20000; i++) {               // (a query costs about a microsecond: at most ~20 ms of polling)
            const hipError_t e = hipStreamQuery(stream);
            if (e == hipSuccess) {
                if (waited) (void)hipGetLastError();    // ("not ready" is an answer, not an error to be found by the next check)
                return hipSuccess;
            }
            if (e != hipErrorNotReady) return e;
            waited = true;
        }
        (void)hipGetLastError();
    }
    return hipStreamSynchronize(stream);
}

static inline u32 ceil_div_u32(u64 a, u64 b) { return (u32)((a + b - 1) / b); }
static inline int bit_width_u32(u32 x) { int b = 0; while (x) { b++; x >>= 1; } return b; }

#define LAUNCH(ctx, kernel, grid, ...) LAUNCH_NAMED(ctx, #kernel, kernel, grid, __VA_ARGS__)

#define LAUNCH_NAMED(ctx, name, kernel, grid, ...)                                         \
    do {                                                                                   \
        if (!(ctx).dry) {                                                                  \
            const bool _p = (ctx).prof && (ctx).prof->enabled;                             \
            if (_p) (ctx).prof->begin(name, (ctx).stream);                                 \
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), 0, (ctx).stream, __VA_ARGS__); \
            HIP_CHECK(hipGetLastError());                                                  \
            if (_p) (ctx).prof->end((ctx).stream);                                         \
        }                                                                                  \
    } while (0)

// (the same with a workgroup size other than BLOCK)
#define LAUNCH_BLOCK(ctx, kernel, grid, block, ...)                                        \
    do {                                                                                   \
        if (!(ctx).dry) {                                                                  \
            const bool _p = (ctx).prof && (ctx).prof->enabled;                             \
            if (_p) (ctx).prof->begin(#kernel, (ctx).stream);                              \
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (ctx).stream, __VA_ARGS__); \
            HIP_CHECK(hipGetLastError());                                                  \
            if (_p) (ctx).prof->end((ctx).stream);                                         \
        }                                                                                  \
    } while (0)

// ---------------------------------------------------------- device utils ----
__device__ __forceinline__ u64 load_u64_unaligned(const uint8_t *p)
{
    u64 x;
    __builtin_memcpy(&x, p, 8);
    return x;
}

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ u32 wave_id() { return threadIdx.x >> 6; }

// __syncthreads() behind LDS atomics that return nothing (ds_add_u32), for use at the top of a loop:
// hipcc (ROCm 7.2, gfx950) was seen to leave out the s_waitcnt lgkmcnt(0) in front of the s_barrier on the
// loop's back edge, so other waves read the counters before this wave's adds had landed.  The wait is
// written as inline asm, which the compiler's wait-count pass neither sees nor removes.
__device__ __forceinline__ void syncthreads_after_lds_atomics()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}

// The budget of deep LCP comparisons of one build (see LCP_SOFT_CAP): LCP_BUDGET_SLOTS counters, zeroed with the build's
// flag words; a workgroup counts in slot blockIdx mod 64.  slots == nullptr: no budget (every comparison may go deep).
__device__ __forceinline__ bool lcp_deep_allowed(const LcpBudget &b)
{
    if (!b.slots) return true;
    u32 *c = b.slots + (blockIdx.x & (LCP_BUDGET_SLOTS - 1u));
    // (a plain look first: once the budget is spent nobody adds any more -- millions of atomics on 64 words would cost
    // what they are meant to save; a stale value only lets a few more comparisons through)
    if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= b.per_slot) return false;
    return atomicAdd(c, 1u) < b.per_slot;
}

// Common prefix of the suffixes at i and j of the byte stream, the first h symbols known to agree (0xFF bytes --
// terminators, each a symbol of its own -- end it; the stream is padded so that 8-byte reads stay in bounds).  Returns the
// exact length, or LCP_PARTIAL_BIT | (a length known to agree) where the comparison was cut: at LCP_SOFT_CAP once the
// build's budget of deep comparisons is spent, at LCP_DIRECT_CAP in any case -- lcp_finish_kernel (tables.h) takes over.
__device__ __forceinline__ u32 lcp_bytes_capped(const uint8_t *__restrict__ s8, u32 i, u32 j, u32 h, const LcpBudget &b)
{
    auto step_of = [](u64 x, u64 y) -> u32 {
        const u64 d = x ^ y, z = ~x;                                 // zero byte of z <=> 0xFF in x
        const u64 t = (z - 0x0101010101010101ull) & ~z & 0x8080808080808080ull;
        const u32 mism = d ? (u32)__builtin_ctzll(d) >> 3 : 8u;
        const u32 term = t ? (u32)__builtin_ctzll(t) >> 3 : 8u;
        return mism < term ? mism : term;
    };
    // (most comparisons end within a few symbols: two steps of 8; what goes on takes 32 symbols per step, eight loads in
    // flight -- a step costs a memory round trip whatever its width.  The byte stream is readable 32 bytes past any
    // position a comparison can reach: it stops at the document's last terminator at the latest, and the arena goes on)
    for (int k = 0; k < 2 && h < LCP_SOFT_CAP; k++) {
        const u32 step = step_of(load_u64_unaligned(s8 + i + h), load_u64_unaligned(s8 + j + h));
        h += step;
        if (step < 8u) return h;
    }
    bool deep = false;
    while (true) {
        if (h >= LCP_SOFT_CAP && !deep) {
            if (!lcp_deep_allowed(b)) return LCP_PARTIAL_BIT | h;
            deep = true;
        }
        if (h >= LCP_DIRECT_CAP) return LCP_PARTIAL_BIT | h;
        u64 x[4], y[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { x[k] = load_u64_unaligned(s8 + i + h + 8 * k); y[k] = load_u64_unaligned(s8 + j + h + 8 * k); }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 step = step_of(x[k], y[k]);
            h += step;
            if (step < 8u) return h;
        }
    }
}

// One flag word for a whole launch ("some comparison was cut short"): on repetitive text nearly every thread raises it, and
// ten million atomics on one word took 1.7 ms of a 10 ms build -- a thread looks first (the lanes of a wavefront share the request).
__device__ __forceinline__ void raise_flag(u32 *flag)
{
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(flag, 1u);
}

// Inclusive prefix sum across the 64 lanes of a wavefront.
__device__ __forceinline__ u32 wave_inclusive_sum(u32 x)
{
    const u32 lane = lane_id();
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        u32 y = __shfl_up(x, off, WAVE);
        if (lane >= (u32)off) x += y;
    }
    return x;
}

__device__ __forceinline__ u32 wave_min(u32 x)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        u32 y = __shfl_xor(x, off, WAVE);
        x = y < x ? y : x;
    }
    return x;
}

__device__ __forceinline__ u32 wave_sum(u32 x)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, WAVE);
    return x;
}

// Exclusive prefix sum of one value per thread over a 256-thread workgroup.
// lds4 must hold 4 words; `total` receives the workgroup sum.
__device__ __forceinline__ u32 block_exclusive_sum(u32 x, u32 *lds4, u32 &total)
{
    const u32 inc = wave_inclusive_sum(x);
    const u32 w = wave_id();
    if (lane_id() == 63) lds4[w] = inc;
    __syncthreads();
    const u32 w0 = lds4[0], w1 = lds4[1], w2 = lds4[2], w3 = lds4[3];
    u32 base = 0;
    if (w > 0) base += w0;
    if (w > 1) base += w1;
    if (w > 2) base += w2;
    total = w0 + w1 + w2 + w3;
    __syncthreads();
    return base + inc - x;
}
