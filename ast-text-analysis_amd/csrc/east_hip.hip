// east_hip.hip -- C ABI (include/east_hip.h) + host orchestration of the
// gfx950 enhanced-annotated-suffix-array build and the score table.
//
// Device data layout (all in one arena, one HIP stream per handle):
//   s        u32[n+3]  dense symbol codes of the concatenated corpus: text code
//                      points ranked 1..sigma_t in code-point order, the g-th
//                      terminator of the corpus = sigma_t+1+g, three 0 pads
//   sa       u32[n]    suffix array, partitioned by document (document d owns
//                      ranks [doc_off[d], doc_off[d+1])), values = global positions
//   lcp/ann  u32[n]    per-document LCP and annotation tables, same layout
//   up/down/next u32[n] child tables, values local to the document
//   s8       u8[n+16]  the same stream, one byte per symbol (0xFF = terminator), when sigma_t <= 254
//   pyramid            16-ary min pyramid over lcp (tables.h)
//
// One suffix sort covers the whole shard: terminators are numbered globally
// (order inside a document preserved, every terminator above every text
// symbol), so stable-partitioning the global suffix array by document yields
// each document's own reference suftab bit for bit.
#include "common.h"
#include "dc3.h"
#include "radix_sort.h"
#include "scan.h"
#include "score.h"
#include "tables.h"
#include "textprep.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include <algorithm>
#include <string.h>

#define TEXT_SYMBOLS EAST_HIP_TERMINATOR_START      // 0x0A00 = 2560 possible text code points
#define PRESENT_WORDS (TEXT_SYMBOLS / 32)           // 80
#define TERM_TAG EAST_HIP_TERMINATOR_TAG            // tagged encoding: bit 31 marks a terminator, text is any code point
#define HI_SYMBOLS (0x110000u - TEXT_SYMBOLS)       // text code points at or above the reference's terminator base
#define HI_WORDS (HI_SYMBOLS / 32u)                 // 34736

static thread_local std::string g_last_error;

// ------------------------------------------------------------ prep kernels --
// (vec: the caller's symbol array is 16-byte aligned, as every allocation is; a misaligned view of a
// larger buffer takes the symbol-by-symbol path)
__global__ __launch_bounds__(BLOCK) void presence_kernel(const u32 *__restrict__ sym, u32 n, int vec,
                                                         u32 *__restrict__ present)
{
    __shared__ u32 bits[PRESENT_WORDS];
    if (threadIdx.x < PRESENT_WORDS) bits[threadIdx.x] = 0;
    __syncthreads();
    // a plain LDS read filters the (overwhelmingly common) already-set case; a stale read only
    // costs a redundant atomic
    auto mark = [&](u32 c) {
        if (c < TEXT_SYMBOLS && !(((volatile u32 *)bits)[c >> 5] & (1u << (c & 31u))))
            atomicOr(&bits[c >> 5], 1u << (c & 31u));
    };
    const u32 stride = gridDim.x * BLOCK;
    const u32 n4 = vec ? n >> 2 : 0u;                      // four symbols per 16-byte load, four loads in flight
    u32 i = blockIdx.x * BLOCK + threadIdx.x;
    for (; i + 3u * stride < n4; i += 4u * stride) {
        const uint4 a = reinterpret_cast<const uint4 *>(sym)[i], b = reinterpret_cast<const uint4 *>(sym)[i + stride];
        const uint4 c = reinterpret_cast<const uint4 *>(sym)[i + 2u * stride], d = reinterpret_cast<const uint4 *>(sym)[i + 3u * stride];
        mark(a.x); mark(a.y); mark(a.z); mark(a.w);
        mark(b.x); mark(b.y); mark(b.z); mark(b.w);
        mark(c.x); mark(c.y); mark(c.z); mark(c.w);
        mark(d.x); mark(d.y); mark(d.z); mark(d.w);
    }
    for (; i < n4; i += stride) {
        const uint4 c = reinterpret_cast<const uint4 *>(sym)[i];
        mark(c.x); mark(c.y); mark(c.z); mark(c.w);
    }
    for (u32 i = (n4 << 2) + blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) mark(sym[i]);
    __syncthreads();
    if (threadIdx.x < PRESENT_WORDS && bits[threadIdx.x]) atomicOr(&present[threadIdx.x], bits[threadIdx.x]);
}

// Dense codes of the text alphabet from the presence bitmap: code point c -> its rank (1 ..) among the code
// points present, 0 if absent; flags[FLAG_SIGMA] = their number.  One workgroup, ten code points per thread.
// `assumed` != NONE: the host went ahead with that alphabet size (speculative build); a different one, or a
// document that does not end in a terminator (doc_status), is recorded in flags[FLAG_STATUS].
#define FLAG_CAPPED 0
#define FLAG_STATUS 1
#define FLAG_SIGMA 2
#define FLAG_KEEP 3
#define FLAG_FAIL 4
#define FLAG_SIGMA_HI 5
#define FLAG_PLACE_FAIL 6        // the placement pass's "a repeat too long to order directly" (window_sort.h: fail), zeroed with the flags
#define FLAG_KG_BAD 7            // the fused finish could not mark every k-gram bucket start (Ctx::kg_bad)
#define FLAG_SAMPLE 8             // 17 words: sample_prefix_kernel's counts (dup2, dup4 for l = 1 .. 8) and the sample size
#define FLAG_WORDS 32
#define STATUS_NO_TERMINATOR 1u
#define STATUS_N_STRINGS 2u
#define STATUS_SIGMA_GUESS 4u
#define STATUS_BAD_SYMBOL 8u
// guess: the code map (TEXT_SYMBOLS words) and the presence bitmap (PRESENT_WORDS words) of the handle's last build,
// kept outside the arena.  A speculative build has already turned the symbols into bytes with that map
// (presence_remap_kernel); `check` then compares the bitmaps -- any difference, and the bytes are wrong: STATUS_SIGMA_GUESS --,
// and in any case the guess is replaced by what this build found.
__global__ __launch_bounds__(BLOCK) void codemap_kernel(const u32 *__restrict__ present, u32 assumed,
                                                        u32 *__restrict__ code_map, u32 *__restrict__ flags,
                                                        u32 *__restrict__ guess = nullptr, int check = 0)
{
    if (guess && threadIdx.x < PRESENT_WORDS) {
        u32 *gp = guess + TEXT_SYMBOLS;
        if (check && gp[threadIdx.x] != present[threadIdx.x]) atomicOr(&flags[FLAG_STATUS], STATUS_SIGMA_GUESS);
        gp[threadIdx.x] = present[threadIdx.x];
    }
    static_assert(TEXT_SYMBOLS == BLOCK * 10, "ten code points per thread");
    __shared__ u32 lds4[WAVES_PER_BLOCK];
    const u32 c0 = threadIdx.x * 10u;
    u32 bits = 0, cnt = 0;
#pragma unroll
    for (u32 i = 0; i < 10; i++) {
        const u32 b = (present[(c0 + i) >> 5] >> ((c0 + i) & 31u)) & 1u;
        bits |= b << i;
        cnt += b;
    }
    u32 total;
    u32 run = block_exclusive_sum(cnt, lds4, total);
#pragma unroll
    for (u32 i = 0; i < 10; i++) {
        const u32 code = ((bits >> i) & 1u) ? ++run : 0u;
        code_map[c0 + i] = code;
        if (guess) guess[c0 + i] = code;
    }
    if (threadIdx.x == 0) {
        flags[FLAG_SIGMA] = total;
        u32 st = present[PRESENT_WORDS] & STATUS_NO_TERMINATOR;
        if (assumed != 0xFFFFFFFFu && assumed != total) st |= STATUS_SIGMA_GUESS;
        if (st) atomicOr(&flags[FLAG_STATUS], st);
    }
}

// Planning sample ("The plan of a first build"): SAMPLE_N consecutive suffixes of the raw symbol stream;
// for every prefix length l = 1 .. SAMPLE_MAX_L, how many of them share their first l symbols (no terminator among
// them) with at least one / at least three others of the sample.  Natural-language text repeats words within a few
// thousand characters, random text does not: the host takes the window width and the fused finish from these counts
// in the same build, no history needed.  One workgroup per prefix length; counts by hashing into a table of 16-bit
// counters in LDS, twice with different hashes, the smaller count taken (count-min).
#define SAMPLE_N 8192
#define SAMPLE_MAX_L 8
#define SAMPLE_SLOTS 16384
#define SAMPLE_THREADS 1024
__global__ __launch_bounds__(SAMPLE_THREADS) void sample_prefix_kernel(const u32 *__restrict__ sym, u32 n, u32 pos0, u32 count,
                                                                       u32 *__restrict__ out)
{
    constexpr int PER = SAMPLE_N / SAMPLE_THREADS;
    __shared__ uint16_t s16[SAMPLE_N + SAMPLE_MAX_L];   // the symbols; 0xFFFF = a terminator (or the end of the stream)
    __shared__ u32 tab[SAMPLE_SLOTS / 2];               // 16-bit counters, two to a word
    __shared__ u32 acc[2 * SAMPLE_MAX_L];
    for (u32 i = threadIdx.x; i < SAMPLE_N + SAMPLE_MAX_L; i += SAMPLE_THREADS) {
        const u32 p = pos0 + i;
        const u32 c = p < n ? sym[p] : 0xFFFFFFFFu;
        s16[i] = c < TEXT_SYMBOLS ? (uint16_t)c : (uint16_t)0xFFFFu;
    }
    if (threadIdx.x < 2 * SAMPLE_MAX_L) acc[threadIdx.x] = 0;
    {
        const int l = (int)blockIdx.x + 1;              // one workgroup per prefix length, side by side
        u32 first_count[PER];
        u32 d2 = 0, d4 = 0;
        for (int variant = 0; variant < 2; variant++) {
            __syncthreads();
            for (u32 i = threadIdx.x; i < SAMPLE_SLOTS / 2; i += SAMPLE_THREADS) tab[i] = 0;
            __syncthreads();
            u32 slot[PER];
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const u32 p = threadIdx.x + SAMPLE_THREADS * q;
                u32 h = 0x811C9DC5u;
                bool ok = p < count;
                for (int t = 0; t < l; t++) {
                    const u32 c = s16[p + t];
                    ok = ok && c != 0xFFFFu;
                    h = (h ^ c) * 0x01000193u;
                }
                h ^= h >> 15;
                h *= variant ? 0x9E3779B1u : 0x85EBCA6Bu;
                slot[q] = ok ? h >> 18 : 0xFFFFFFFFu;   // 14 bits
                if (ok) atomicAdd(&tab[slot[q] >> 1], 1u << (16u * (slot[q] & 1u)));
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const u32 c = slot[q] != 0xFFFFFFFFu ? (tab[slot[q] >> 1] >> (16u * (slot[q] & 1u))) & 0xFFFFu : 0u;
                if (variant == 0) first_count[q] = c;
                else {
                    const u32 both = c < first_count[q] ? c : first_count[q];
                    d2 += both >= 2u ? 1u : 0u;
                    d4 += both >= 4u ? 1u : 0u;
                }
            }
        }
        d2 = wave_sum(d2);
        d4 = wave_sum(d4);
        if (lane_id() == 0) { atomicAdd(&acc[2 * (l - 1)], d2); atomicAdd(&acc[2 * (l - 1) + 1], d4); }
    }
    __syncthreads();
    if (threadIdx.x < 2) out[2 * blockIdx.x + threadIdx.x] = acc[2 * blockIdx.x + threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[2 * SAMPLE_MAX_L] = count;
}

// Speculative build: presence bitmap AND byte stream in one pass over the symbols, the bytes through the code map of
// the handle's last build (guess; codemap_kernel finds out whether that was right).  16-byte aligned input only.
__global__ __launch_bounds__(BLOCK) void presence_remap_kernel(const u32 *__restrict__ sym, u32 n,
                                                               const u32 *__restrict__ guess, u32 *__restrict__ present,
                                                               uint8_t *__restrict__ s8)
{
    __shared__ u32 bits[PRESENT_WORDS];
    __shared__ __attribute__((aligned(16))) uint8_t map8[TEXT_SYMBOLS];
    if (threadIdx.x < PRESENT_WORDS) bits[threadIdx.x] = 0;
    {
        // (the code map's 2 560 words with 16-byte loads, all of a thread's requested before the first is used -- own
        // allocation, 256-byte aligned --: a loop of one word per step was ten round trips in front of the first symbol)
        static_assert(TEXT_SYMBOLS % 4 == 0, "whole 16-byte groups");
        constexpr u32 GROUPS = TEXT_SYMBOLS / 4, ROUNDS = (GROUPS + BLOCK - 1) / BLOCK;
        uint4 q[ROUNDS];
#pragma unroll
        for (u32 r = 0; r < ROUNDS; r++) {
            const u32 g = threadIdx.x + r * BLOCK;
            q[r] = reinterpret_cast<const uint4 *>(guess)[g < GROUPS ? g : 0u];
        }
#pragma unroll
        for (u32 r = 0; r < ROUNDS; r++) {
            const u32 g = threadIdx.x + r * BLOCK;
            if (g < GROUPS)
                reinterpret_cast<u32 *>(map8)[g] = (q[r].x & 0xFFu) | ((q[r].y & 0xFFu) << 8) | ((q[r].z & 0xFFu) << 16) | ((q[r].w & 0xFFu) << 24);
        }
    }
    __syncthreads();
    auto code = [&](u32 c) -> u32 {
        if (c >= TEXT_SYMBOLS) return 0xFFu;
        if (!(((volatile u32 *)bits)[c >> 5] & (1u << (c & 31u)))) atomicOr(&bits[c >> 5], 1u << (c & 31u));
        return map8[c];
    };
    auto word = [&](const uint4 q) -> u32 { return code(q.x) | (code(q.y) << 8) | (code(q.z) << 16) | (code(q.w) << 24); };
    const u32 stride = gridDim.x * BLOCK, n4 = n >> 2;
    u32 *out = reinterpret_cast<u32 *>(s8);
    u32 i = blockIdx.x * BLOCK + threadIdx.x;
    for (; i + 3u * stride < n4; i += 4u * stride) {       // four 16-byte loads in flight
        const uint4 a = reinterpret_cast<const uint4 *>(sym)[i], b = reinterpret_cast<const uint4 *>(sym)[i + stride];
        const uint4 c = reinterpret_cast<const uint4 *>(sym)[i + 2u * stride], d = reinterpret_cast<const uint4 *>(sym)[i + 3u * stride];
        out[i] = word(a);
        out[i + stride] = word(b);
        out[i + 2u * stride] = word(c);
        out[i + 3u * stride] = word(d);
    }
    for (; i < n4; i += stride) out[i] = word(reinterpret_cast<const uint4 *>(sym)[i]);
    if (blockIdx.x == 0 && threadIdx.x < 20u) {            // the last n % 4 symbols and the 16 pad bytes
        const u32 j = (n4 << 2) + threadIdx.x;
        if (j < n) s8[j] = (uint8_t)code(sym[j]);
        else if (j < n + 16u) s8[j] = 0;
    }
    __syncthreads();
    if (threadIdx.x < PRESENT_WORDS && bits[threadIdx.x]) atomicOr(&present[threadIdx.x], bits[threadIdx.x]);
}

struct TermIn {                                 // 1 at terminators; defined on [0, n]
    const u32 *sym;
    u32 n;
    u32 tagged;
    __device__ __forceinline__ u32 operator()(u32 i) const
    {
        return (i < n && (tagged ? sym[i] >> 31 : (u32)(sym[i] >= TEXT_SYMBOLS))) ? 1u : 0u;
    }
};

// ---- tagged encoding: text code points at or above U+0A00 ------------------------------------------
// (a stream whose terminators carry EAST_HIP_TERMINATOR_TAG may hold any code point as text; the kernels above
// and below read such a stream unchanged as long as no text symbol reaches U+0A00 -- a tagged terminator is
// ">= U+0A00" --, and that is found out here)
// (the whole bitmap -- 136 KiB -- lives in the workgroup's LDS: a plain LDS read filters the bits that are set, which
// in CJK text is every symbol after the first few thousand; one 1024-thread workgroup per CU, grid-stride, and only
// the words a workgroup has touched go to the global bitmap)
#define PRESENCE_HI_THREADS 1024
__global__ __launch_bounds__(PRESENCE_HI_THREADS) void presence_hi_kernel(const u32 *__restrict__ sym, u32 n,
                                                                          u32 *__restrict__ hi_bits, u32 *__restrict__ status)
{
    __shared__ u32 bits[HI_WORDS];
    for (u32 w = threadIdx.x; w < HI_WORDS; w += PRESENCE_HI_THREADS) bits[w] = 0;
    __syncthreads();
    const u32 stride = gridDim.x * PRESENCE_HI_THREADS;
    for (u32 i = blockIdx.x * PRESENCE_HI_THREADS + threadIdx.x; i < n; i += stride) {
        const u32 c = sym[i];
        if (c < TEXT_SYMBOLS || (c >> 31)) continue;
        if (c >= 0x110000u) { atomicOr(status, STATUS_BAD_SYMBOL); continue; }
        const u32 k = c - TEXT_SYMBOLS;
        if (!(((volatile u32 *)bits)[k >> 5] & (1u << (k & 31u)))) atomicOr(&bits[k >> 5], 1u << (k & 31u));
    }
    __syncthreads();
    for (u32 w = threadIdx.x; w < HI_WORDS; w += PRESENCE_HI_THREADS)
        if (bits[w]) atomicOr(&hi_bits[w], bits[w]);
}

// hi_rank[w] = code points present below word w of the bitmap; flags[FLAG_SIGMA_HI] = their number.  One workgroup.
__global__ __launch_bounds__(BLOCK) void hi_rank_kernel(const u32 *__restrict__ hi_bits, u32 *__restrict__ hi_rank,
                                                        u32 *__restrict__ flags)
{
    __shared__ u32 lds4[WAVES_PER_BLOCK];
    const u32 per = (HI_WORDS + BLOCK - 1) / BLOCK;
    const u32 w0 = threadIdx.x * per, w1 = w0 + per < HI_WORDS ? w0 + per : HI_WORDS;
    u32 cnt = 0;
    for (u32 w = w0; w < w1; w++) cnt += __popc(hi_bits[w]);
    u32 total;
    u32 run = block_exclusive_sum(cnt, lds4, total);
    for (u32 w = w0; w < w1; w++) { hi_rank[w] = run; run += __popc(hi_bits[w]); }
    if (threadIdx.x == 0) flags[FLAG_SIGMA_HI] = total;
}

__device__ __forceinline__ u32 hi_rank_of(const u32 *__restrict__ hi_bits, const u32 *__restrict__ hi_rank, u32 k)
{
    return hi_rank[k >> 5] + __popc(hi_bits[k >> 5] & ((1u << (k & 31u)) - 1u));
}

// dense codes when text at or above U+0A00 is present (tagged encoding): ONE text alphabet in code-point order --
// code points below U+0A00 through the code map (1..sigma_lo), those above by their rank in the bitmap
// (sigma_lo+1 ..) --, the terminators above it as ever.  s8 / s: the byte stream (sigma_t <= 254) or the u32 codes.
__global__ __launch_bounds__(BLOCK) void remap_hi_kernel(const u32 *__restrict__ sym, const u32 *__restrict__ term_ex,
                                                         const u32 *__restrict__ code_map,
                                                         const u32 *__restrict__ hi_bits, const u32 *__restrict__ hi_rank,
                                                         u32 sigma_lo, u32 sigma_t, u32 n, u32 *__restrict__ s,
                                                         uint8_t *__restrict__ s8)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) {
        const u32 c = sym[i];
        const bool term = c >> 31;
        u32 code;
        if (term) code = s ? sigma_t + 1u + term_ex[i] : 0xFFu;
        else if (c < TEXT_SYMBOLS) code = code_map[c];
        else code = sigma_lo + 1u + hi_rank_of(hi_bits, hi_rank, c < 0x110000u ? c - TEXT_SYMBOLS : 0u);
        if (s) s[i] = code; else s8[i] = (uint8_t)code;
    } else {
        if (s && i < n + 3) s[i] = 0;
        if (s8 && i < n + 16) s8[i] = 0;
    }
}

// byte path: only the byte stream is built (0xFF = terminator); the exact terminator numbers are
// never needed there, so no terminator scan runs
__global__ __launch_bounds__(BLOCK) void remap_bytes_kernel(const u32 *__restrict__ sym,
                                                            const u32 *__restrict__ code_map, u32 n, int vec,
                                                            uint8_t *__restrict__ s8)
{
    // sixteen symbols per thread: four 16-byte loads in flight, one 16-byte store (the grid covers n + 16 bytes)
    const u32 i = (blockIdx.x * BLOCK + threadIdx.x) * 16u;
    if (vec && i + 16u <= n) {
        uint4 c[4];
#pragma unroll
        for (int q = 0; q < 4; q++) c[q] = reinterpret_cast<const uint4 *>(sym + i)[q];
        u32 out[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u32 b0 = c[q].x < TEXT_SYMBOLS ? code_map[c[q].x] & 0xFFu : 0xFFu, b1 = c[q].y < TEXT_SYMBOLS ? code_map[c[q].y] & 0xFFu : 0xFFu;
            const u32 b2 = c[q].z < TEXT_SYMBOLS ? code_map[c[q].z] & 0xFFu : 0xFFu, b3 = c[q].w < TEXT_SYMBOLS ? code_map[c[q].w] & 0xFFu : 0xFFu;
            out[q] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        }
        *reinterpret_cast<uint4 *>(s8 + i) = uint4{out[0], out[1], out[2], out[3]};
    } else {
        for (u32 j = i; j < i + 16u && j < n + 16u; j++) {
            uint8_t v = 0;
            if (j < n) { const u32 c = sym[j]; v = c < TEXT_SYMBOLS ? (uint8_t)code_map[c] : (uint8_t)0xFF; }
            s8[j] = v;
        }
    }
}

// memory safety: every document must end in a terminator (it stops every suffix comparison)
__global__ __launch_bounds__(BLOCK) void validate_last_symbol_kernel(const u32 *__restrict__ sym,
                                                                     const u32 *__restrict__ doc_off, u32 n_docs,
                                                                     u32 tagged, u32 *__restrict__ status)
{
    const u32 d = blockIdx.x * BLOCK + threadIdx.x;
    if (d >= n_docs) return;
    const u32 c = sym[doc_off[d + 1] - 1];
    if (tagged ? !(c >> 31) : c < TEXT_SYMBOLS) atomicOr(status, STATUS_NO_TERMINATOR);
}

// n_strings[d] must equal the terminators of document d.  Terminators sort above every text
// symbol, so the terminator-first suffixes are the tail of the document's suffix array: one
// search per document on the finished array replaces a counting pass over the corpus.  A wavefront
// per document, 64 probes per step: a 64 MiB document takes 5 steps of two dependent loads instead of 26.
template <class SYM>
__global__ __launch_bounds__(BLOCK) void validate_n_strings_kernel(const SYM *__restrict__ s, u32 term_first,
                                                                   const u32 *__restrict__ sa,
                                                                   const u32 *__restrict__ doc_off,
                                                                   const u32 *__restrict__ n_strings, u32 n_docs,
                                                                   u32 *__restrict__ status)
{
    const u32 d = blockIdx.x * WAVES_PER_BLOCK + wave_id(), lane = lane_id();
    if (d >= n_docs) return;
    u32 lo = doc_off[d], hi = doc_off[d + 1];           // the first terminator-first rank lies in [lo, hi]
    const u32 end = hi, last = doc_off[n_docs] - 1u;
    while (lo < hi) {
        const u32 step = (hi - lo + 63u) / 64u;
        const u64 probe = (u64)lo + (u64)lane * step;
        bool term = true;                               // (probes at or behind hi count as terminator-first)
        if (probe < hi) {
            const u32 q = sa[(u32)probe];
            const u32 p = q < last ? q : last;          // (a speculative build that guessed wrong leaves stale entries)
            term = (u32)s[p] >= term_first;
        }
        const u64 bal = __ballot(term);
        const u32 t = bal ? (u32)__ffsll((unsigned long long)bal) - 1u : 64u;
        const u32 new_hi = t < 64u ? (u32)((u64)lo + (u64)t * step < hi ? (u64)lo + (u64)t * step : hi) : hi;
        const u32 new_lo = t > 0u ? lo + (t - 1u) * step + 1u : lo;
        lo = new_lo < new_hi ? new_lo : new_hi;
        hi = new_hi;
    }
    if (lane == 0 && end - lo != n_strings[d]) atomicOr(status, STATUS_N_STRINGS);
}

__global__ __launch_bounds__(BLOCK) void remap_kernel(const u32 *__restrict__ sym,
                                                      const u32 *__restrict__ term_ex,
                                                      const u32 *__restrict__ code_map, u32 sigma_t,
                                                      u32 n, u32 *__restrict__ s, uint8_t *__restrict__ s8)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) {
        const u32 c = sym[i];
        const bool text = c < TEXT_SYMBOLS;
        const u32 code = text ? code_map[c] : sigma_t + 1u + term_ex[i];
        s[i] = code;
        if (s8) s8[i] = text ? (uint8_t)code : (uint8_t)0xFF;
    } else {
        if (i < n + 3) s[i] = 0;
        if (s8 && i < n + 16) s8[i] = 0;
    }
}

// ------------------------------------------------------------------ index --
struct east_hip_index {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    Arena arena;
    Stats stats;
    Profiler prof;
    bool built = false, child_built = false;
    u32 n = 0, n_docs = 0, sigma_t = 0, m_total = 0;
    u32 sigma_hi = 0;            // text code points >= U+0A00 present (tagged encoding only): the top sigma_hi codes of the text alphabet
    u32 *hi_bits = nullptr, *hi_rank = nullptr;      // presence bitmap over [U+0A00, U+110000) and its rank directory
    u32 *guess = nullptr;        // code map + presence bitmap of the last build (own allocation): what a speculative build starts from
    bool tagged_input = false;   // east_hip_set_symbol_encoding: the symbol entry points take the tagged encoding
    bool prep_tagged = false;    // the prepared symbols (east_hip_build_texts) are in the tagged encoding
    int bits0 = 0;
    std::vector<i64> h_doc_off;
    std::vector<u32> h_n_strings;
    // persistent device arrays (inside the arena)
    uint8_t *s8 = nullptr;
    bool use_s8 = false;
    u32 *s = nullptr, *sa = nullptr, *lcp = nullptr, *ann = nullptr, *up = nullptr, *down = nullptr,
        *next = nullptr, *doc_off = nullptr, *n_strings = nullptr, *code_map = nullptr;
    Pyramid pyr;
    u32 build_docs = 0;          // documents of the build in progress (h->n_docs is set when it has succeeded)
    // what the last successful build found, the guesses of the next (speculative) one
    bool hint_valid = false, hint_no_rounds = false, hint_window = false;
    int plan_wide = -1, plan_fused = -1;   // what the last build's window sort did (wide first window, fused finish): a speculative build does the same
    int plan_ht = -1;                      // ... (first-level keys of variable-length code words)
    int plan_persist = -1;                 // ... (the first domain went straight to the persistent rounds)
    // the order-preserving variable-length code of the last build that made one (ht_code.h): device tables (own
    // allocation: 256 x u32 enc, 4096 x u16 dec), valid for text with ht_sigma text symbols
    char *ht_tab = nullptr;
    bool ht_valid = false;
    u32 ht_sigma = 0;
    int ht_max_len = 0;
    double ht_mean_len = 0.0;
    u32 hint_sigma = 0;
    u32 plan_n = 0, plan_docs = 0, plan_epoch = 0;
    bool plan_tagged = false;   // shape of the last sizing run (and test-knob epoch), its result
    size_t plan_bytes = 0;
    // keyphrases + score scratch (own allocation, grown on demand)
    char *q_buf = nullptr;
    size_t q_cap = 0;
    u32 n_kp = 0, n_q = 0, score_chunk = 0;     // score_chunk: documents per stretch the scratch was sized for
    u32 *q_raw = nullptr, *q_code = nullptr, *q_end = nullptr, *q_off = nullptr, *group_off = nullptr;
    u32 *q_blk = nullptr;        // whole keyphrases per workgroup of the score walk: [q_blk[i], q_blk[i + 1]), n_blk of them
    u32 n_blk = 0;               // (0: a keyphrase is longer than a workgroup -- the walk writes per-suffix results, a second kernel sums)
    double *suffix = nullptr, *table = nullptr, *table_g = nullptr;
    // k-gram bucket tables for the score walk (own allocation, rebuilt after every build)
    u32 *kg = nullptr;
    size_t kg_cap = 0;
    int kg_k = 0;
    u32 kg_A = 0, kg_bins = 0;
    bool kg_built = false;
    bool kg_marked = false;      // the bucket starts were written by the build (off the window keys): only the fill is due
    bool kg_pairs = false;       // ... in the pair layout (score.h: KgTables); kg3 = the table of the levels above the last,
    u32 *kg3 = nullptr, *kg_up = nullptr;        // kg_up = the small tables of the levels above kg3's own
    u32 kg_up_stride = 0;
    float last_build_ms = -1.f, last_score_ms = -1.f, last_prep_ms = -1.f;
    // the caller's Unicode tables of the device text preparation (own allocation, re-uploaded when their hash changes)
    char *tp_tables = nullptr;
    size_t tp_tables_bytes = 0;
    u64 tp_tables_hash = 0;
    std::vector<uint8_t> tp_host_tables;
    // the streamed text preparation: a copy stream of its own and one event per chunk
    hipStream_t copy_stream = nullptr;     // (created with the handle: creating a stream costs milliseconds)
    std::vector<hipEvent_t> copy_events;
    // ... and a ring of pinned host memory through which MANY separate texts go up (tp_upload_through_ring): allocated on
    // first use, kept with the handle
    char *ring = nullptr;
    std::vector<hipEvent_t> ring_events;
    std::thread ring_alloc;                 // pins the ring in the background after a first call that went without it
    std::atomic<char *> ring_pending{nullptr};
    std::atomic<bool> ring_done{false};     // the background thread is through (with or without a ring): it can be joined without waiting
    int narrow_upload = 0;                  // the last build's host symbols went up as 16-bit words (1) / as bytes (2) (east_hip_build_info [25])
    bool bytes_refused = false;             // a text symbol of 0xFF .. 0x9FF was met on the way up as bytes: this handle's later uploads take 16 bits at once
    bool ring_wanted = false;               // (the call under way would have taken the ring: pin it once the call is over --
                                            // while it runs, the pinning and the call's own copies fight over the runtime's locks)
    // symbols prepared on the device by east_hip_build_texts (own allocation)
    u32 *prep_sym = nullptr;
    size_t prep_cap = 0;
    i64 prep_n = 0;
    std::vector<i64> prep_doc_off;
    std::vector<int32_t> prep_n_strings;
};

struct SpecAbort {};             // a speculative build cannot go on: build_common starts over with the read-backs in place

// Every entry point runs on the handle's device and puts the calling thread's current device back on the way
// out (a caller that also drives torch or other HIP code on another GPU must not find its device changed).
static thread_local int g_saved_device = -1;
static void use_device_ordinal(int device)
{
    int cur = -1;
    if (g_saved_device < 0 && hipGetDevice(&cur) == hipSuccess && cur != device) g_saved_device = cur;
    HIP_CHECK(hipSetDevice(device));
}
static void use_device(east_hip_index *h) { use_device_ordinal(h->device); }
static void restore_device()
{
    if (g_saved_device >= 0) { (void)hipSetDevice(g_saved_device); g_saved_device = -1; }
}

static bool kgram_reserve(east_hip_index *h, u64 bins, u32 n_docs);

// min pyramid over the LCP table and the annotation table: the streaming pass decides all but the widest
// intervals and writes pyramid level 1 on the way, the upper levels follow, then the listed wide ones
static void annotate(east_hip_index *h, Ctx &ctx)
{
    const Pyramid &pyr = h->pyr;
    const u32 n = pyr.len[0];
    Arena &ar = *ctx.arena;
    const size_t mark = ar.mark();
    const u32 n_tiles = ceil_div_u32(n, ANN_TILE);
    u32 *wide_list = ar.alloc<u32>((size_t)n_tiles * ANN_TILE);       // every tile has its own stretch
    u32 *wide_count = ar.alloc<u32>(n_tiles);
    const bool has_lvl1 = pyr.levels > 1;                // (a table of at most 16 entries has no level 1)
    LAUNCH(ctx, ann_stream_kernel, n_tiles, pyr.ptr[0], (const u32 *)h->doc_off, (const u32 *)h->n_strings,
           h->build_docs, n, h->ann, has_lvl1 ? (u32 *)pyr.ptr[1] : (u32 *)nullptr, has_lvl1 ? pyr.len[1] : 0u,
           has_lvl1 ? pyr_padded(pyr.len[1]) : 0u, wide_list, wide_count, h->lcp);
    // the levels above: one launch each while they are large, the top of the pyramid in a single one
    int l = 2;
    for (; l < pyr.levels && pyr.len[l] > PYR_TOP; l++)
        LAUNCH(ctx, pyramid_level_kernel, ceil_div_u32(pyr_padded(pyr.len[l]), BLOCK), pyr.ptr[l - 1], pyr.len[l],
               pyr_padded(pyr.len[l]), (u32 *)pyr.ptr[l]);
    if (l < pyr.levels) LAUNCH(ctx, pyramid_top_kernel, 1, pyr, l);
    LAUNCH(ctx, ann_wide_kernel, ceil_div_u32(n_tiles, BLOCK / ANN_WIDE_SLOTS), pyr, n, n_tiles, (const u32 *)wide_list,
           (const u32 *)wide_count, h->ann);
    ar.release(mark);
}

// symbol counts of the byte stream (the weights of the variable-length code, ht_code.h): 16 bytes per thread and step,
// per-lane-class counters in LDS
__global__ __launch_bounds__(BLOCK) void byte_hist_kernel(const uint8_t *__restrict__ s8, u32 n, u32 *__restrict__ counts)
{
    __shared__ u32 bins[4][256];
    for (int c = 0; c < 4; c++) bins[c][threadIdx.x] = 0;
    __syncthreads();
    u32 *mine = bins[threadIdx.x & 3u];
    for (u64 i = ((u64)blockIdx.x * BLOCK + threadIdx.x) * 16u; i < n; i += (u64)gridDim.x * BLOCK * 16u) {
        const uint4 x = *reinterpret_cast<const uint4 *>(s8 + i);          // (the stream is padded to 16 bytes behind n)
        const u32 wds[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int q = 0; q < 16; q++)
            if (i + q < n) atomicAdd(&mine[(wds[q >> 2] >> ((q & 3) * 8)) & 0xFFu], 1u);
    }
    __syncthreads();
    const u32 t = bins[0][threadIdx.x] + bins[1][threadIdx.x] + bins[2][threadIdx.x] + bins[3][threadIdx.x];
    if (t) atomicAdd(&counts[threadIdx.x], t);
}

// The variable-length code of this build's text (ht_code.h), for the window sort to use if it pays: made from the symbol
// counts on a build that waits for the device anyway, taken over from the build before on a speculative one (same
// alphabet -- any alphabetic code orders correctly, a stale one is merely less compact).
static void prepare_ht_code(east_hip_index *h, Ctx &ctx, u32 n, u32 sigma_t, u32 m_total)
{
    ctx.ht_max_len = 0;
    if (ctx.dry || !h->use_s8 || !ctx.knobs.window_sort || ctx.knobs.ht_mode == 0) return;
    if (ctx.spec) {
        if (!h->ht_valid || h->ht_sigma != sigma_t) {
            // (no code from the build before.  Forced -- the tests -- the build starts over with its read-backs in place and makes one)
            if (ctx.knobs.ht_mode == 1 && sigma_t + 1 >= 8 && n >= 64) throw SpecAbort();
            return;
        }
    } else {
        h->ht_valid = false;
        // (an alphabet of at most 5 bits -- letters only -- has nothing to gain: at best a fraction of a symbol per key)
        if (sigma_t + 1 < 8 || (ctx.knobs.ht_mode != 1 && (bit_width_u32(sigma_t + 1) < 6 || n < 65536)) || n < 64) return;
        Arena &ar = *ctx.arena;
        const size_t mark = ar.mark();
        u32 *d_counts = ar.alloc<u32>(256);
        HIP_CHECK(hipMemsetAsync(d_counts, 0, 256 * sizeof(u32), ctx.stream));
        LAUNCH(ctx, byte_hist_kernel, std::min<u32>(ceil_div_u32(n, BLOCK * 16), 2048), (const uint8_t *)h->s8, n, d_counts);
        u32 h_counts[256];
        HIP_CHECK(hipMemcpyAsync(h_counts, d_counts, sizeof(h_counts), hipMemcpyDeviceToHost, ctx.stream));
        HIP_CHECK(sync_stream(ctx.stream));
        ar.release(mark);
        std::vector<u64> counts(256);
        for (int c = 0; c < 256; c++) counts[c] = h_counts[c];
        std::vector<u32> enc;
        std::vector<uint16_t> dec;
        double mean_len = 0.0;
        if (!ht_make_tables(counts, sigma_t, m_total, enc, dec, &mean_len)) return;
        if (!h->ht_tab) {
            void *p = nullptr;
            if (hipMalloc(&p, 256 * 4 + HT_DEC_SIZE * 2) != hipSuccess) { (void)hipGetLastError(); return; }
            h->ht_tab = (char *)p;
        }
        // (pageable host buffers: the copies are staged before the calls return)
        HIP_CHECK(hipMemcpyAsync(h->ht_tab, enc.data(), 256 * 4, hipMemcpyHostToDevice, ctx.stream));
        HIP_CHECK(hipMemcpyAsync(h->ht_tab + 256 * 4, dec.data(), HT_DEC_SIZE * 2, hipMemcpyHostToDevice, ctx.stream));
        HIP_CHECK(sync_stream(ctx.stream));
        int longest = 0;
        for (u32 c = 0; c < 256; c++) longest = std::max(longest, (int)(enc[c] & 0xFFu));
        h->ht_valid = true;
        h->ht_sigma = sigma_t;
        h->ht_max_len = longest;
        h->ht_mean_len = mean_len;
        if (g_trace) fprintf(stderr, "[east_hip] variable-length code: %u symbols, %.2f bits per symbol on average, longest code word %d\n",
                             sigma_t + 1, mean_len, longest);
    }
    ctx.ht_enc = (const u32 *)h->ht_tab;
    ctx.ht_dec = (const uint16_t *)(h->ht_tab + 256 * 4);
    ctx.ht_max_len = h->ht_max_len;
    ctx.ht_mean_len = h->ht_mean_len;
}

// The build proper.  With ctx.dry it only measures the arena high-water mark
// (worst case: widest keys, recursion to the bottom).
static void build_impl(east_hip_index *h, Ctx &ctx, const u32 *d_sym, u32 n, u32 n_docs,
                       const i64 *doc_offsets, const int32_t *n_strings, u32 spec_sigma = 0, bool tagged = false)
{
    Arena &ar = *ctx.arena;
    ar.release(0);
    // ---- persistent arrays --------------------------------------------------
    h->s = ar.alloc<u32>((size_t)n + 3);
    h->s8 = ar.alloc<uint8_t>((size_t)n + 64);           // (16 zero bytes behind the stream; comparisons look up to 32 bytes ahead)
    h->sa = ar.alloc<u32>(n);
    h->lcp = ar.alloc<u32>(pyr_padded(n));
    h->ann = ar.alloc<u32>(n);
    h->up = ar.alloc<u32>(n);
    h->down = ar.alloc<u32>(n);
    h->next = ar.alloc<u32>(n);
    h->doc_off = ar.alloc<u32>((size_t)n_docs + 1);
    h->n_strings = ar.alloc<u32>(n_docs);
    h->code_map = ar.alloc<u32>(TEXT_SYMBOLS + FLAG_WORDS + PRESENT_WORDS + 1 + LCP_BUDGET_SLOTS);   // + the flag words (FLAG_*) + the presence bitmap and its status word + the budget of deep LCP comparisons
    h->hi_bits = tagged ? ar.alloc<u32>(HI_WORDS) : nullptr;
    h->hi_rank = tagged ? ar.alloc<u32>(HI_WORDS) : nullptr;
    Pyramid pyr;
    pyr.levels = 1;
    pyr.ptr[0] = h->lcp;
    pyr.len[0] = n;
    while (pyr.len[pyr.levels - 1] > PYR_FAN) {
        if (pyr.levels >= PYR_MAX_LEVELS) east_throw(EAST_HIP_ERR_INTERNAL, "pyramid too deep");
        const u32 len = ceil_div_u32(pyr.len[pyr.levels - 1], PYR_FAN);
        pyr.ptr[pyr.levels] = ar.alloc<u32>(pyr_padded(len));
        pyr.len[pyr.levels] = len;
        pyr.levels++;
    }
    h->pyr = pyr;
    h->build_docs = n_docs;

    // ---- host-side small tables ---------------------------------------------
    std::vector<u32> off32((size_t)n_docs + 1), m32(n_docs);
    u32 longest_doc = ctx.dry ? n : 0;                    // (sizing run: as if one document held everything)
    if (!ctx.dry) {
        for (u32 d = 0; d <= n_docs; d++) off32[d] = (u32)doc_offsets[d];
        for (u32 d = 0; d < n_docs; d++) longest_doc = std::max(longest_doc, off32[d + 1] - off32[d]);
        for (u32 d = 0; d < n_docs; d++) m32[d] = (u32)n_strings[d];
        HIP_CHECK(hipMemcpyAsync(h->doc_off, off32.data(), off32.size() * 4, hipMemcpyHostToDevice, ctx.stream));
        HIP_CHECK(hipMemcpyAsync(h->n_strings, m32.data(), m32.size() * 4, hipMemcpyHostToDevice, ctx.stream));
    }

    const u32 gn = ceil_div_u32(n, BLOCK);
    u32 sigma_t = TEXT_SYMBOLS - 1, m_total = n;          // dry-run worst case
    u32 sigma_hi = tagged ? HI_SYMBOLS : 0;
    u32 *flags = h->code_map + TEXT_SYMBOLS;              // flag words behind the code map
    u32 *capped = flags + FLAG_CAPPED, *status = flags + FLAG_STATUS;
    u32 *present = flags + FLAG_WORDS;                    // PRESENT_WORDS + 1 (status word): zeroed in the same fill
    if (!ctx.dry) HIP_CHECK(hipMemsetAsync(flags, 0, (FLAG_WORDS + PRESENT_WORDS + 1 + LCP_BUDGET_SLOTS) * sizeof(u32), ctx.stream));
    // (deep LCP comparisons -- beyond LCP_SOFT_CAP symbols -- this build may make: n / 256, see common.h)
    ctx.lcp_budget.slots = ctx.dry ? nullptr : present + PRESENT_WORDS + 1;
    ctx.lcp_budget.per_slot = std::max<u32>(n / 256u / LCP_BUDGET_SLOTS, 4u);
    ctx.spec_out = flags + FLAG_KEEP;
    ctx.zeroed_word = ctx.dry ? nullptr : flags + FLAG_PLACE_FAIL;
    ctx.kg_bad = ctx.dry ? nullptr : flags + FLAG_KG_BAD;
    {
        // ---- alphabet, dense remap ---------------------------------------------
        const size_t mark = ar.mark();
        u32 *term_ex = ar.alloc<u32>((size_t)n + 1);      // wide-alphabet path only
        const int vec = ((uintptr_t)d_sym & 15u) == 0;
        const bool fused = ctx.spec && vec && h->guess;      // (the bytes come out of the same pass, through the last build's map)
        if (fused) LAUNCH(ctx, presence_remap_kernel, std::min<u32>(gn, 2048), d_sym, n, (const u32 *)h->guess, present, h->s8);
        else LAUNCH(ctx, presence_kernel, std::min<u32>(gn, 2048), d_sym, n, vec, present);
        LAUNCH(ctx, validate_last_symbol_kernel, ceil_div_u32(n_docs, BLOCK), d_sym, (const u32 *)h->doc_off, n_docs,
               (u32)tagged, present + PRESENT_WORDS);
        if (tagged) {
            if (!ctx.dry) HIP_CHECK(hipMemsetAsync(h->hi_bits, 0, HI_WORDS * 4, ctx.stream));
            LAUNCH_BLOCK(ctx, presence_hi_kernel, std::min<u32>(ceil_div_u32(n, PRESENCE_HI_THREADS), 256), PRESENCE_HI_THREADS,
                         d_sym, n, h->hi_bits, status);
            LAUNCH(ctx, hi_rank_kernel, 1, (const u32 *)h->hi_bits, h->hi_rank, flags);
        }
        LAUNCH(ctx, codemap_kernel, 1, (const u32 *)present, ctx.spec ? spec_sigma : 0xFFFFFFFFu, h->code_map, flags, h->guess,
               (int)fused);
        ctx.sample_n = 0;
        ctx.rep_n = ctx.rep_dup = 0;
        bool sample_plans = false;                          // the sample is large enough to plan the window and the fused finish from
        if (!ctx.dry && !ctx.spec && !tagged && n >= 1024u) {
            // the planning sample: the middle of the longest document (read back together with the alphabet)
            u32 dl = 0;
            for (u32 d = 1; d < n_docs; d++)
                if (off32[d + 1] - off32[d] > off32[dl + 1] - off32[dl]) dl = d;
            const u32 len = off32[dl + 1] - off32[dl], cnt = std::min<u32>(SAMPLE_N, len);
            // (many short documents: a sample of a few dozen suffixes decides nothing -- no sample, the uniform estimates.
            // A small input's sample -- 512 suffixes or more -- only answers "is this text repetitive?": the reference's
            // worst-case collection at n = 300, 30 K symbols, took 3.7 ms through the endgame's direct ordering of its
            // groups of 100 and takes 0.6 through the persistent rounds, window_sort.h)
            sample_plans = n >= 4 * SAMPLE_N && cnt >= SAMPLE_N / 4;
            if (cnt >= 512u)
                LAUNCH_BLOCK(ctx, sample_prefix_kernel, SAMPLE_MAX_L, SAMPLE_THREADS, d_sym, n, off32[dl] + (len - cnt) / 2, cnt,
                             flags + FLAG_SAMPLE);
        }
        if (!ctx.dry) {
            if (ctx.spec) {
                sigma_t = spec_sigma;                        // (checked on the device; found out at the end of the build)
            } else {
                u32 hf[FLAG_WORDS];
                HIP_CHECK(hipMemcpyAsync(hf, flags, sizeof(hf), hipMemcpyDeviceToHost, ctx.stream));
                HIP_CHECK(sync_stream(ctx.stream));
                if (hf[FLAG_STATUS] & STATUS_NO_TERMINATOR)
                    east_throw(EAST_HIP_ERR_DOMAIN, tagged ? "a document does not end in a (tagged) string terminator"
                                                           : "a document does not end in a string terminator (>= U+0A00)");
                if (hf[FLAG_STATUS] & STATUS_BAD_SYMBOL)
                    east_throw(EAST_HIP_ERR_DOMAIN, "a symbol is neither a tagged terminator nor a code point < U+110000");
                sigma_t = hf[FLAG_SIGMA];
                sigma_hi = tagged ? hf[FLAG_SIGMA_HI] : 0;
                ctx.sample_n = hf[FLAG_SAMPLE + 2 * SAMPLE_MAX_L];
                for (int l = 1; l <= SAMPLE_MAX_L; l++) {
                    ctx.sample_dup2[l] = hf[FLAG_SAMPLE + 2 * (l - 1)];
                    ctx.sample_dup4[l] = hf[FLAG_SAMPLE + 2 * (l - 1) + 1];
                }
                ctx.rep_n = ctx.sample_n;
                ctx.rep_dup = ctx.sample_dup4[SAMPLE_MAX_L];
                if (!sample_plans) ctx.sample_n = 0;          // (too small to plan from)
                if (g_trace && ctx.sample_n) {
                    fprintf(stderr, "[east_hip] sample of %u suffixes, shared prefixes (>= 2 / >= 4 of the sample) by length:", ctx.sample_n);
                    for (int l = 1; l <= SAMPLE_MAX_L; l++) fprintf(stderr, " %d: %u/%u", l, ctx.sample_dup2[l], ctx.sample_dup4[l]);
                    fprintf(stderr, "\n");
                }
            }
            m_total = 0;
            for (u32 d = 0; d < n_docs; d++) m_total += (u32)n_strings[d];     // checked after the build
            h->use_s8 = sigma_t + sigma_hi <= 254;
        }
        sigma_t += sigma_hi;                                  // one text alphabet, the high code points above the low ones
        if (sigma_hi && h->use_s8 && !ctx.dry) {
            LAUNCH(ctx, remap_hi_kernel, ceil_div_u32((u64)n + 16, BLOCK), d_sym, (const u32 *)nullptr,
                   (const u32 *)h->code_map, (const u32 *)h->hi_bits, (const u32 *)h->hi_rank, sigma_t - sigma_hi, sigma_t, n,
                   (u32 *)nullptr, h->s8);
        } else if (fused) {
            // (done)
        } else if (h->use_s8 && !ctx.dry) {
            LAUNCH(ctx, remap_bytes_kernel, ceil_div_u32((u64)n + 16, BLOCK * 16), d_sym, (const u32 *)h->code_map, n,
                   vec, h->s8);
        } else {
            // wide alphabets: dense u32 codes, terminators numbered globally by a scan
            device_scan<TermIn, false>(ctx, TermIn{d_sym, n, (u32)tagged}, n + 1, term_ex);
            if (sigma_hi)
                LAUNCH(ctx, remap_hi_kernel, ceil_div_u32((u64)n + 16, BLOCK), d_sym, (const u32 *)term_ex,
                       (const u32 *)h->code_map, (const u32 *)h->hi_bits, (const u32 *)h->hi_rank, sigma_t - sigma_hi, sigma_t, n,
                       h->s, (uint8_t *)nullptr);
            else
                LAUNCH(ctx, remap_kernel, ceil_div_u32((u64)n + 16, BLOCK), d_sym, (const u32 *)term_ex,
                       (const u32 *)h->code_map, sigma_t, n, h->s, (uint8_t *)nullptr);
        }
        ar.release(mark);
    }
    const u32 sigma = sigma_t + m_total;
    const u32 term_first = sigma_t + 1u;
    h->sigma_hi = ctx.dry ? 0u : sigma_hi;
    h->sigma_t = sigma_t;
    h->m_total = m_total;
    h->bits0 = bit_width_u32(sigma);

    // ---- suffix array of the whole shard, then partition by document -------------
    // Text first goes through the window sort over all suffixes -- with several documents the keys carry
    // the document number on top, so that every document's tables come out side by side --; DC3 is the
    // bounded-work fallback (several documents: one suffix sort of the whole shard, then a stable
    // partition by document).
    bool window_sorted = false;
    const int doc_bits = n_docs > 1 ? bit_width_u32(n_docs - 1) : 0;
    h->kg_marked = false;
    prepare_ht_code(h, ctx, n, sigma_t, m_total);
    if ((h->use_s8 || ctx.dry) && ctx.knobs.window_sort) {       // (the sizing run prices it with 64-bit keys)
        DocKey docs;
        if (n_docs > 1) { docs.doc_off = h->doc_off; docs.n_docs = n_docs; docs.bits = doc_bits; docs.h_doc_off = ctx.dry ? nullptr : off32.data(); }
        // the score walk's k-gram tables are marked off the sorted keys on the way (KgMark): as many levels
        // as a table of at most twice a document's size (and 1 GiB in all) has room for
        KgMark km;
        if (!ctx.dry && n_docs <= 65535 && !getenv("EAST_HIP_NO_KG_MARKS")) {     // (the variable: experiments -- the score side then builds its tables itself)
            km.A = sigma_t + 2;
            u64 bins = 1;
            while (km.k < KGRAM_KEYS_MAX_K && bins * km.A <= KGRAM_KEYS_MAX_BINS && bins * km.A <= 2 * ((u64)n / n_docs) + 4096 &&
                   (bins * km.A + 1) * n_docs * 8 <= ((u64)1 << 31)) {
                bins *= km.A;
                km.k++;
            }
            if (km.k > 0 && kgram_reserve(h, bins, n_docs)) {
                km.kg = h->kg;
                km.doc_off = h->doc_off;
                km.n_docs = n_docs;
                // (the pair layout: the level above the last in a table of its own, behind the 8-byte entries)
                // (... where the score table has many columns: with a handful of long documents the 8-byte marks cost the
                // build more than the few thousand walks of a score call get back; forced by the test knob)
                if (ctx.knobs.kg_pairs && (n_docs >= 16 || ctx.knobs.kg_pairs_forced)) km.kg3 = h->kg + 2 * (size_t)(bins + 1) * n_docs;
                h->kg3 = km.kg3;
                h->kg_up = km.kg3 ? km.kg3 + (size_t)(bins / km.A + 1) * n_docs : nullptr;
            } else {
                km.k = 0;
            }
        }
        window_sorted = window_suffix_sort(ctx, h->s8, n, sigma_t + 1, h->sa, h->lcp, capped, docs, longest_doc,
                                           km.k > 0 ? &km : nullptr);
        if (window_sorted && km.k > 0) {                 // (km.k is 0 if the sort did not mark: small inputs)
            h->kg_marked = true;
            h->kg_pairs = km.pairs != 0;
            h->kg_k = km.k;
            h->kg_A = km.A;
            h->kg_bins = km.bins;
        }
    }
    ctx.stats->window_sorted = window_sorted;
    if (ctx.spec && !window_sorted) throw SpecAbort();   // (DC3 is not written to survive a wrong guess of the alphabet)
    // on the byte stream the LCP table comes with the suffix array: from the window keys, or (one
    // document) out of the level-0 merge of DC3
    const bool fused_lcp = h->use_s8 && (window_sorted || n_docs == 1);
    if (window_sorted) {
        ctx.stats->levels = 0;
    } else if (n_docs == 1) {
        ctx.stats->levels = dc3_suffix_array(ctx, h->s, n, sigma, h->sa, 0, term_first, h->use_s8 ? h->s8 : nullptr,
                                             fused_lcp ? h->lcp : nullptr, capped);
    } else {
        // the suffix array of the whole shard lands in vals[0]; the partition is a stable radix sort of
        // (document, suffix) pairs whose last pass writes into h->sa (pass i reads buffers [i % 2], writes the others)
        const size_t mark_sa = ar.mark();
        SortBufs<u32> sb;
        const int last = radix_pass_count(doc_bits) & 1;
        for (int k = 0; k < 2; k++) { sb.keys[k] = ar.alloc<u32>((size_t)n + 4); sb.vals[k] = k == last ? h->sa : ar.alloc<u32>((size_t)n + 4); }
        u32 *sa_whole = sb.vals[0];
        ctx.stats->levels = dc3_suffix_array(ctx, h->s, n, sigma, sa_whole, 0, term_first, h->use_s8 ? h->s8 : nullptr);
        if (n_docs <= DOC_LDS_MAX) {
            LAUNCH(ctx, doc_keys_lds_kernel, ceil_div_u32(n, BLOCK * 4), (const u32 *)sa_whole, (const u32 *)h->doc_off, n_docs,
                   n, sb.keys[0]);
        } else {
            const int shift = std::max(0, bit_width_u32(n) - 20);
            const u32 n_coarse = (u32)(((u64)n >> shift) + 1);
            u32 *coarse = ar.alloc<u32>(n_coarse);
            LAUNCH(ctx, doc_coarse_kernel, ceil_div_u32(n_coarse, BLOCK), (const u32 *)h->doc_off, n_docs, n, shift,
                   n_coarse, coarse);
            LAUNCH(ctx, doc_keys_kernel, ceil_div_u32(n, BLOCK * 4), (const u32 *)sa_whole, (const u32 *)h->doc_off,
                   (const u32 *)coarse, shift, n, sb.keys[0]);
        }
        const int r = radix_sort_pairs<u32>(ctx, sb, n, doc_bits);
        if (sb.vals[r] != h->sa) east_throw(EAST_HIP_ERR_INTERNAL, "document partition ended in the wrong buffer");
        ar.release(mark_sa);
    }

    // n_strings against the terminators actually present (read back at the end of the build)
    if (h->use_s8)
        LAUNCH_NAMED(ctx, "validate_n_strings_kernel", (validate_n_strings_kernel<uint8_t>), ceil_div_u32(n_docs, WAVES_PER_BLOCK),
                     (const uint8_t *)h->s8, 0xFFu, (const u32 *)h->sa, (const u32 *)h->doc_off,
                     (const u32 *)h->n_strings, n_docs, status);
    else
        LAUNCH_NAMED(ctx, "validate_n_strings_kernel", (validate_n_strings_kernel<u32>), ceil_div_u32(n_docs, WAVES_PER_BLOCK),
                     (const u32 *)h->s, sigma_t + 1u, (const u32 *)h->sa, (const u32 *)h->doc_off,
                     (const u32 *)h->n_strings, n_docs, status);

    // ---- LCP, min pyramid, annotation + child tables ------------------------------
    {
        const size_t mark = ar.mark();
        u32 *rank = ctx.dry ? ar.alloc<u32>(n) : nullptr;       // only allocated for real when needed
        if (fused_lcp) {
            // (the table's padding up to a multiple of 16 is written by ann_stream_kernel)
        } else if (h->use_s8) {
            LAUNCH(ctx, lcp8_kernel, ceil_div_u32(pyr_padded(n), BLOCK), (const uint8_t *)h->s8, (const u32 *)h->sa,
                   n, h->lcp, capped, ctx.lcp_budget);
        } else {
            LAUNCH(ctx, lcp_kernel, ceil_div_u32(pyr_padded(n), BLOCK), (const u32 *)h->s, (const u32 *)h->sa,
                   n, h->lcp, capped, ctx.lcp_budget);
        }
        if (n_docs > 1)
            LAUNCH(ctx, lcp_doc_starts_kernel, ceil_div_u32(n_docs, BLOCK), (const u32 *)h->doc_off, n_docs, h->lcp);
        (void)rank;
        ar.release(mark);
    }
    // (comparisons that hit the cap -- a repetitive input -- are found out about at the end of the build,
    // together with the status word: build_common then finishes those ranks and redoes the two steps below)
    annotate(h, ctx);
    h->kg_built = false;
    h->child_built = false;      // childtab_up / down / next_l_index: built on first east_hip_get_tables request
}

// A repetitive input: the ranks whose direct comparison was capped are finished with the Kasai carry
// over the text (blocked, O(n + marked work)), then pyramid and annotation are built again.
static void finish_capped_lcp(east_hip_index *h, Ctx &ctx)
{
    Arena &ar = *ctx.arena;
    const u32 n = h->pyr.len[0];
    const size_t mark = ar.mark();
    const u32 gn = ceil_div_u32(n, BLOCK);
    u32 *rank = ar.alloc<u32>(n), *count = ar.alloc<u32>(n), *apos = ar.alloc<u32>(n);
    u32 *list = ar.alloc<u32>(n), *long_list = ar.alloc<u32>(n), *counters = ar.alloc<u32>(2);
    uint8_t *anchor = ar.alloc<uint8_t>((size_t)n + 16);
    const void *sym = h->use_s8 ? (const void *)h->s8 : (const void *)h->s;
    HIP_CHECK(hipMemsetAsync(counters, 0, 2 * sizeof(u32), ctx.stream));
    LAUNCH(ctx, inverse_sa_kernel, gn, (const u32 *)h->sa, n, rank);
    // (tables.h, "finishing pass for unfinished LCP entries": classify, compare the irreducible positions, fill the rest)
    if (h->use_s8) {
        LAUNCH(ctx, (lcp_phi_classify_kernel<true>), gn, sym, (const u32 *)h->sa, (const u32 *)rank, (const u32 *)h->lcp, n, anchor, list, counters);
        LAUNCH(ctx, (lcp_phi_compare_kernel<true>), std::min<u32>(ceil_div_u32(n, WAVES_PER_BLOCK), 8192u), sym, (const u32 *)h->sa, (const u32 *)rank, (const u32 *)list,
               (const u32 *)counters, n, h->lcp, long_list, counters + 1);
        LAUNCH_BLOCK(ctx, (lcp_phi_long_kernel<true>), 512, PHI_LONG_THREADS, sym, (const u32 *)h->sa, (const u32 *)rank, (const u32 *)long_list,
                     (const u32 *)(counters + 1), n, h->lcp);
    } else {
        LAUNCH(ctx, (lcp_phi_classify_kernel<false>), gn, sym, (const u32 *)h->sa, (const u32 *)rank, (const u32 *)h->lcp, n, anchor, list, counters);
        LAUNCH(ctx, (lcp_phi_compare_kernel<false>), std::min<u32>(ceil_div_u32(n, WAVES_PER_BLOCK), 8192u), sym, (const u32 *)h->sa, (const u32 *)rank, (const u32 *)list,
               (const u32 *)counters, n, h->lcp, long_list, counters + 1);
        LAUNCH_BLOCK(ctx, (lcp_phi_long_kernel<false>), 512, PHI_LONG_THREADS, sym, (const u32 *)h->sa, (const u32 *)rank, (const u32 *)long_list,
                     (const u32 *)(counters + 1), n, h->lcp);
    }
    device_scan<U8In, true>(ctx, U8In{anchor}, n, count);
    LAUNCH(ctx, lcp_phi_anchors_kernel, gn, (const uint8_t *)anchor, (const u32 *)count, n, apos);
    LAUNCH(ctx, lcp_phi_fill_kernel, gn, (const uint8_t *)anchor, (const u32 *)count, (const u32 *)apos, (const u32 *)rank, n, h->lcp);
    ar.release(mark);
    annotate(h, ctx);
}

static size_t plan_arena_bytes(u32 n, u32 n_docs, bool lean = false, bool tagged = false, const Knobs *knobs = nullptr)
{
    // (a tagged stream is priced both ways: as the byte stream it becomes when its alphabet is small -- the window
    // sort with its rounds, which the sizing run of the widest alphabet never enters -- and as dense u32 codes)
    size_t high = 0;
    for (int t = 0; t <= (tagged ? 1 : 0); t++) {
        east_hip_index tmp;
        Arena dry;
        dry.dry = true;
        Stats st;
        Ctx ctx;
        if (knobs) ctx.knobs = *knobs;
        ctx.arena = &dry;
        ctx.dry = true;
        ctx.lean = lean;
        ctx.stats = &st;
        build_impl(&tmp, ctx, nullptr, n, n_docs, nullptr, nullptr, 0, t == 1);
        high = std::max(high, dry.high + (tagged ? 2 * (size_t)HI_WORDS * 4 + 1024 : 0));
    }
    return high + (1u << 20);
}

static void ensure_arena(east_hip_index *h, size_t bytes)
{
    if (h->arena.cap >= bytes) return;
    HIP_CHECK(hipStreamSynchronize(h->stream));
    if (h->arena.base) HIP_CHECK(hipFree(h->arena.base));
    h->arena.base = nullptr;
    h->arena.cap = 0;
    h->built = false;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        char b[160];
        snprintf(b, sizeof(b), "hipMalloc of the %.2f GiB build arena failed: %s", bytes / 1073741824.0,
                 hipGetErrorString(e));
        east_throw(EAST_HIP_ERR_OOM, b);
    }
    h->arena.base = (char *)p;
    h->arena.cap = bytes;
}

static void check_build_args(i64 n_total, const i64 *doc_offsets, const int32_t *n_strings, int32_t n_docs)
{
    if (n_docs < 1 || !doc_offsets || !n_strings) east_throw(EAST_HIP_ERR_INVALID, "n_docs < 1 or null offsets");
    if (n_total < 1 || n_total >= (i64)0x7FFFFFF0) east_throw(EAST_HIP_ERR_INVALID, "n_total must be in [1, 2^31-16)");
    if (doc_offsets[0] != 0 || doc_offsets[n_docs] != n_total)
        east_throw(EAST_HIP_ERR_INVALID, "doc_offsets must start at 0 and end at n_total");
    for (int32_t d = 0; d < n_docs; d++) {
        if (doc_offsets[d + 1] <= doc_offsets[d]) east_throw(EAST_HIP_ERR_INVALID, "empty document (the reference raises EmptyStringsCollectionException)");
        if (n_strings[d] < 1 || n_strings[d] > doc_offsets[d + 1] - doc_offsets[d])
            east_throw(EAST_HIP_ERR_INVALID, "n_strings[d] must be in [1, n_d]");
    }
}

// ---- host symbols go up as 16-bit words ------------------------------------------------------------------------------
// east_hip_build hands over 4 bytes per symbol, and the link moves 56 GB/s: 245 MB for the 64 MiB bench document are 4.4 ms
// before the 1.7 ms build can start (tools/pcie_probe.py: pageable and pinned memory alike).  In the reference's encoding
// a text symbol is below U+0A00 and everything else a terminator whose number the build never reads, so half the bytes
// say it all: host threads narrow the symbols into the slots of the pinned ring (0xFFFF = "a terminator"), the slots go
// up one DMA each, and a kernel behind every DMA widens them again into the staging area the build reads -- the link
// carries 2 bytes per symbol, narrowing and widening hide under it.  (Tagged streams -- text above U+0A00 -- and small
// inputs take the plain copy; a handle's first call too, while the ring is pinned in the background.)
#define TP_RING_SLOTS 3                     // slots of TP_RING_SLOT bytes in a handle's pinned ring (common.h; tp_fill_stream further down)
#define SYM_NARROW_MIN ((u32)4 << 20)
#define SYM_TERMINATOR16 0xFFFFu
static void ring_pin_later(east_hip_index *h);
// The ring a background thread pinned becomes the handle's (or is given back when the handle pinned one itself in the
// meantime); wait: join the thread even if it is still at work (before an inline allocation, at destruction).
static void ring_adopt(east_hip_index *h, bool wait)
{
    if (h->ring_alloc.joinable() && (wait || h->ring_done.load())) h->ring_alloc.join();
    if (h->ring_alloc.joinable()) return;
    char *pending = h->ring_pending.exchange(nullptr);
    if (!pending) return;
    if (!h->ring) h->ring = pending;
    else if (pending != h->ring) (void)hipHostFree(pending);
}
__global__ __launch_bounds__(BLOCK) void widen_symbols_kernel(const uint16_t *__restrict__ in, u32 n, u32 *__restrict__ out)
{
    const u32 i = (blockIdx.x * BLOCK + threadIdx.x) * 8u;
    if (i + 8u <= n && ((uintptr_t)(in + i) & 15u) == 0 && ((uintptr_t)(out + i) & 15u) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(in + i);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
        u32 o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u32 x = (w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
            o[k] = x == SYM_TERMINATOR16 ? TEXT_SYMBOLS : x;
        }
        reinterpret_cast<uint4 *>(out + i)[0] = uint4{o[0], o[1], o[2], o[3]};
        reinterpret_cast<uint4 *>(out + i)[1] = uint4{o[4], o[5], o[6], o[7]};
    } else {
        for (u32 j = i; j < i + 8u && j < n; j++) { const u32 x = in[j]; out[j] = x == SYM_TERMINATOR16 ? TEXT_SYMBOLS : x; }
    }
}

// (bytes: text code points below 0xFF as they are, 0xFF = a terminator)
__global__ __launch_bounds__(BLOCK) void widen_symbols8_kernel(const uint8_t *__restrict__ in, u32 n, u32 *__restrict__ out)
{
    const u32 i = (blockIdx.x * BLOCK + threadIdx.x) * 16u;
    if (i + 16u <= n && ((uintptr_t)(in + i) & 15u) == 0 && ((uintptr_t)(out + i) & 15u) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(in + i);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int g = 0; g < 4; g++) {
            u32 o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32 x = (w[g] >> (k * 8)) & 0xFFu;
                o[k] = x == 0xFFu ? TEXT_SYMBOLS : x;
            }
            reinterpret_cast<uint4 *>(out + i)[g] = uint4{o[0], o[1], o[2], o[3]};
        }
    } else {
        for (u32 j = i; j < i + 16u && j < n; j++) { const u32 x = in[j]; out[j] = x == 0xFFu ? TEXT_SYMBOLS : x; }
    }
}

// symbols [0, n) from the host into `staging` (device, n words) through the pinned ring; d_narrow: n + 8 halfwords of device scratch
// The narrowing of a stretch of host symbols into the pinned ring.  With AVX2 (looked for at run time): sixteen symbols a
// step -- unsigned compare by max, saturating pack, the lanes put back in order -- and STREAMING stores: the ring is
// written once and read by the copy engine, a store that first fetches the line it overwrites moves a third more bytes
// through the host's memory than the narrowing needs (4 B read + 2 B written per symbol).
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(__i386__))
#include <immintrin.h>
__attribute__((target("avx2"))) static void narrow_symbols_avx2(const u32 *src, uint16_t *dst, size_t n)
{
    size_t i = 0;
    for (; i < n && ((uintptr_t)(dst + i) & 31u); i++) dst[i] = src[i] < TEXT_SYMBOLS ? (uint16_t)src[i] : (uint16_t)SYM_TERMINATOR16;
    const __m256i first_term = _mm256_set1_epi32((int)TEXT_SYMBOLS), term = _mm256_set1_epi32((int)SYM_TERMINATOR16);
    for (; i + 16 <= n; i += 16) {
        __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
        __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 8));
        const __m256i ta = _mm256_cmpeq_epi32(_mm256_max_epu32(a, first_term), a);      // a >= TEXT_SYMBOLS (unsigned)
        const __m256i tb = _mm256_cmpeq_epi32(_mm256_max_epu32(b, first_term), b);
        a = _mm256_blendv_epi8(a, term, ta);
        b = _mm256_blendv_epi8(b, term, tb);
        const __m256i p = _mm256_permute4x64_epi64(_mm256_packus_epi32(a, b), 0xD8);    // (the pack works per 128-bit lane)
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), p);
    }
    for (; i < n; i++) dst[i] = src[i] < TEXT_SYMBOLS ? (uint16_t)src[i] : (uint16_t)SYM_TERMINATOR16;
    _mm_sfence();
}
static const bool g_have_avx2 = __builtin_cpu_supports("avx2") && getenv("EAST_HIP_NO_AVX2") == nullptr;
#else
static void narrow_symbols_avx2(const u32 *, uint16_t *, size_t) {}
static const bool g_have_avx2 = false;
#endif
// ... and to BYTES, for text whose code points all lie below 0xFF (every BASELINE input: A-Z): half the bytes over the link
// again.  A text symbol the byte cannot hold (0xFF .. 0x9FF) is reported and the upload starts over with 16-bit words.
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(__i386__))
__attribute__((target("avx2"))) static bool narrow_symbols8_avx2(const u32 *src, uint8_t *dst, size_t n)
{
    size_t i = 0;
    bool bad = false;
    auto one = [&](size_t k) { const u32 c = src[k]; bad |= c >= 0xFFu && c < TEXT_SYMBOLS; dst[k] = c < 0xFFu ? (uint8_t)c : (uint8_t)0xFFu; };
    for (; i < n && ((uintptr_t)(dst + i) & 31u); i++) one(i);
    const __m256i first_term = _mm256_set1_epi32((int)TEXT_SYMBOLS), byte_max = _mm256_set1_epi32(0xFF);
    __m256i wrong = _mm256_setzero_si256();
    for (; i + 32 <= n; i += 32) {
        __m256i v[4];
#pragma GCC unroll 4
        for (int k = 0; k < 4; k++) {
            const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 8 * k));
            const __m256i is_term = _mm256_cmpeq_epi32(_mm256_max_epu32(a, first_term), a);           // a >= TEXT_SYMBOLS (unsigned)
            const __m256i fits = _mm256_cmpeq_epi32(_mm256_min_epu32(a, byte_max), a);                  // a <= 0xFF
            // (0xFF itself does not fit either: it is the terminator's byte)
            wrong = _mm256_or_si256(wrong, _mm256_andnot_si256(is_term, _mm256_or_si256(_mm256_cmpeq_epi32(a, byte_max),
                                                                                         _mm256_xor_si256(fits, _mm256_set1_epi32(-1)))));
            v[k] = _mm256_blendv_epi8(a, byte_max, is_term);
        }
        // 32-bit -> 16-bit -> 8-bit, the 128-bit lanes put back in order at the end
        const __m256i p01 = _mm256_packus_epi32(v[0], v[1]), p23 = _mm256_packus_epi32(v[2], v[3]);
        const __m256i b = _mm256_packus_epi16(p01, p23);
        const __m256i r = _mm256_permutevar8x32_epi32(b, _mm256_setr_epi32(0, 4, 1, 5, 2, 6, 3, 7));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), r);
    }
    bad |= !_mm256_testz_si256(wrong, wrong);
    for (; i < n; i++) one(i);
    _mm_sfence();
    return !bad;
}
#else
static bool narrow_symbols8_avx2(const u32 *, uint8_t *, size_t) { return false; }
#endif
static bool narrow_symbols8(const u32 *src, uint8_t *dst, size_t n)
{
    if (g_have_avx2) return narrow_symbols8_avx2(src, dst, n);
    bool bad = false;
    for (size_t i = 0; i < n; i++) { const u32 c = src[i]; bad |= c >= 0xFFu && c < TEXT_SYMBOLS; dst[i] = c < 0xFFu ? (uint8_t)c : (uint8_t)0xFFu; }
    return !bad;
}
static void narrow_symbols(const u32 *src, uint16_t *dst, size_t n)
{
    if (g_have_avx2) { narrow_symbols_avx2(src, dst, n); return; }
    for (size_t i = 0; i < n; i++) dst[i] = src[i] < TEXT_SYMBOLS ? (uint16_t)src[i] : (uint16_t)SYM_TERMINATOR16;
}

// T = uint16_t: every symbol of the reference encoding fits (a terminator = 0xFFFF on the wire); T = uint8_t: text below
// 0xFF only -- returns false, with nothing left in flight, when a symbol did not fit (the caller starts over with 16 bits).
template <class T>
static bool upload_symbols_narrow(east_hip_index *h, const u32 *sym, u32 n, u32 *staging, T *d_narrow)
{
    constexpr bool BYTES = sizeof(T) == 1;
    static const size_t slot_env = getenv("EAST_HIP_SYMBOL_SLOT") ? (size_t)atoll(getenv("EAST_HIP_SYMBOL_SLOT")) : 0;     // (experiments)
    const size_t slot_bytes = slot_env >= 65536 && slot_env <= TP_RING_SLOT ? slot_env & ~(size_t)255 : TP_RING_SLOT;
    const size_t slot_syms = slot_bytes / sizeof(T);
    const u32 n_slots = ceil_div_u32(n, slot_syms);
    static const int threads_env = getenv("EAST_HIP_SYMBOL_THREADS") ? atoi(getenv("EAST_HIP_SYMBOL_THREADS")) : 0;     // (experiments)
    // (bytes: the link carries a quarter of the symbols' bytes, the narrowing threads read all of them -- eight, measured below)
    const int n_fill = threads_env > 0 ? std::min(threads_env, 64)
                                       : (int)std::min<u32>(BYTES ? 8u : 6u, std::max<u32>(2u, std::thread::hardware_concurrency() / 2u));
    // (measured on the 256-thread host of the MI355X box, 61 M symbols: 3 threads 5.8-6.4 ms per call, 4: 5.2-5.5, 6: 4.7-5.6,
    // 8-24: 4.9-5.9 -- against 6.1 ms with the plain 4-byte copy; the narrowing threads, not the link, set the pace)
    if (h->ring_events.empty())
        for (int i = 0; i < TP_RING_SLOTS; i++) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            h->ring_events.push_back(e);
        }
    // (the copy stream must not start before what is still queued on the handle's stream has left the arena alone)
    HIP_CHECK(hipEventRecord(h->ev0, h->stream));
    HIP_CHECK(hipStreamWaitEvent(h->copy_stream, h->ev0, 0));
    std::vector<std::atomic<int>> slot_parts(n_slots);
    for (auto &a : slot_parts) a.store(0, std::memory_order_relaxed);
    std::atomic<u32> slots_free{TP_RING_SLOTS};
    std::atomic<int> abort{0}, misfit{0};
    T *ring = (T *)h->ring;                                  // (slot k of the ring starts at k * TP_RING_SLOT whatever part of it is used)
    std::vector<std::thread> fillers;
    for (int j = 0; j < n_fill; j++)
        fillers.emplace_back([&, j]() {
            for (u32 sl = 0; sl < n_slots; sl++) {
                while (slots_free.load(std::memory_order_acquire) <= sl) {
                    if (abort.load(std::memory_order_acquire)) return;
                    std::this_thread::yield();
                }
                const size_t a = (size_t)sl * slot_syms, len = std::min<size_t>(slot_syms, (size_t)n - a);
                const size_t lo = len * (size_t)j / (size_t)n_fill, hi = len * (size_t)(j + 1) / (size_t)n_fill;
                const u32 *src = sym + a;
                T *dst = ring + (size_t)(sl % TP_RING_SLOTS) * (TP_RING_SLOT / sizeof(T));
                if constexpr (BYTES) { if (!narrow_symbols8(src + lo, (uint8_t *)dst + lo, hi - lo)) misfit.store(1, std::memory_order_release); }
                else narrow_symbols(src + lo, (uint16_t *)dst + lo, hi - lo);
                slot_parts[sl].fetch_add(1, std::memory_order_release);
            }
        });
    struct Joiner {
        std::vector<std::thread> &fill;
        std::atomic<int> &abort;
        hipStream_t copy;
        bool ok = false;
        ~Joiner()
        {
            if (!ok) abort.store(1, std::memory_order_release);
            for (auto &f : fill)
                if (f.joinable()) f.join();
            if (!ok) (void)hipStreamSynchronize(copy);
        }
    } joiner{fillers, abort, h->copy_stream};
    for (u32 sl = 0; sl < n_slots; sl++) {
        while (slot_parts[sl].load(std::memory_order_acquire) < n_fill) std::this_thread::yield();
        if (misfit.load(std::memory_order_acquire)) return false;      // (the joiner stops the fill threads and drains the copy stream)
        const size_t a = (size_t)sl * slot_syms, len = std::min<size_t>(slot_syms, (size_t)n - a);
        HIP_CHECK(hipMemcpyAsync(d_narrow + a, ring + (size_t)(sl % TP_RING_SLOTS) * (TP_RING_SLOT / sizeof(T)), len * sizeof(T), hipMemcpyHostToDevice, h->copy_stream));
        HIP_CHECK(hipEventRecord(h->ring_events[sl % TP_RING_SLOTS], h->copy_stream));
        if constexpr (BYTES)
            hipLaunchKernelGGL(widen_symbols8_kernel, dim3(ceil_div_u32(len, BLOCK * 16)), dim3(BLOCK), 0, h->copy_stream, (const uint8_t *)(d_narrow + a),
                               (u32)len, staging + a);
        else
            hipLaunchKernelGGL(widen_symbols_kernel, dim3(ceil_div_u32(len, BLOCK * 8)), dim3(BLOCK), 0, h->copy_stream, (const uint16_t *)(d_narrow + a),
                               (u32)len, staging + a);
        HIP_CHECK(hipGetLastError());
        if (sl >= 1) {                                   // the slot before is on the device: back to the fill threads
            HIP_CHECK(hipEventSynchronize(h->ring_events[(sl - 1) % TP_RING_SLOTS]));
            slots_free.store(sl + TP_RING_SLOTS, std::memory_order_release);
        }
    }
    joiner.ok = true;
    HIP_CHECK(hipEventRecord(h->ev1, h->copy_stream));   // (ev0 / ev1 are recorded anew by the build behind this)
    HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev1, 0));
    return true;
}

static void build_common(east_hip_index *h, const u32 *sym, bool sym_on_host, i64 n_total, const i64 *doc_offsets,
                         const int32_t *n_strings, int32_t n_docs, bool tagged)
{
    if (!h) east_throw(EAST_HIP_ERR_INVALID, "null handle");
    if (!sym) east_throw(EAST_HIP_ERR_INVALID, "null symbols");
    check_build_args(n_total, doc_offsets, n_strings, n_docs);
    use_device(h);
    h->built = false;
    const Knobs kn = knobs_snapshot();                   // (the test knobs of this call, from its sizing run to its last launch)
    const u32 n = (u32)n_total;
    // (host symbols of the reference encoding go up as 16-bit words where that pays: upload_symbols_narrow)
    ring_adopt(h, false);
    const bool narrow_shape = sym_on_host && !tagged && n >= SYM_NARROW_MIN && getenv("EAST_HIP_NO_SYMBOL_NARROW") == nullptr;
    if (narrow_shape && !h->ring) {
        // The first host-resident build of this size pins the ring in line (3-5 ms, once per handle) and goes up narrowed
        // already: until round 6 it took the plain copy, left the pinning to a background thread, and the call behind it
        // -- arriving while that thread was still at work -- took the plain copy again (6.9, 6.7, then 3.5 ms a call).
        ring_adopt(h, true);                                // (a background pin under way: its ring)
        if (!h->ring) {
            void *p = nullptr;
            if (hipHostMalloc(&p, TP_RING_SLOT * TP_RING_SLOTS, hipHostMallocDefault) == hipSuccess) h->ring = (char *)p;
            else (void)hipGetLastError();                    // (no ring: the plain copy)
        }
    }
    const bool narrow = narrow_shape && h->ring != nullptr;
    const size_t narrow_bytes = narrow_shape ? (((size_t)n + 8) * 2 + 255) & ~(size_t)255 : 0;    // (also while the ring is still being pinned: the arena is sized once)
    const size_t staging_bytes = (sym_on_host ? ((size_t)n * 4 + 255) & ~(size_t)255 : 0) + narrow_bytes;
    if (h->plan_n != n || h->plan_docs != (u32)n_docs || h->plan_epoch != kn.plan_epoch || h->plan_tagged != tagged) {     // (the sizing run costs host time: remembered per shape)
        h->plan_bytes = plan_arena_bytes(n, (u32)n_docs, false, tagged, &kn);
        h->plan_tagged = tagged;
        h->plan_n = n;
        h->plan_docs = (u32)n_docs;
        h->plan_epoch = kn.plan_epoch;
    }
    size_t need = h->plan_bytes + staging_bytes;
    bool lean = kn.force_lean;
    if (!lean && need > h->arena.cap) {
        // the tie-refinement rounds are the largest consumer: when they do not fit next to what else
        // lives on the device, build without them (heavy ties then take the DC3 recursion)
        size_t free_b = 0, total_b = 0;
        HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        lean = need > (size_t)(0.92 * (double)(free_b + h->arena.cap));
    }
    if (lean) need = plan_arena_bytes(n, (u32)n_docs, true, tagged, &kn) + staging_bytes;
    u32 *staging = nullptr;
    {
        const size_t cap_before = h->arena.cap;
        const auto t_a = std::chrono::steady_clock::now();
        ensure_arena(h, need);
        if (g_trace && h->arena.cap != cap_before)
            fprintf(stderr, "[east_hip] build: arena grown to %.2f GiB in %.2f ms\n", h->arena.cap / 1073741824.0,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_a).count());
    }
    if (sym_on_host) {   // raw symbols are staged at the top of the arena
        staging = (u32 *)(h->arena.base + (h->arena.cap - (((size_t)n * 4 + 255) & ~(size_t)255)));
        int went = 0;
        if (narrow) {
            // bytes first where the text may fit them (the first 64 Ki symbols say: word text does, Cyrillic does not); a
            // symbol that does not fit further on starts the upload over with 16-bit words, for this call and the handle's later ones
            static const bool no_bytes = getenv("EAST_HIP_NO_SYMBOL_BYTES") != nullptr;
            bool bytes = !no_bytes && !h->bytes_refused;
            for (u32 i = 0; bytes && i < std::min<u32>(n, 65536u); i++) bytes = sym[i] < 0xFFu || sym[i] >= TEXT_SYMBOLS;
            if (bytes && upload_symbols_narrow<uint8_t>(h, sym, n, staging, (uint8_t *)((char *)staging - narrow_bytes))) went = 2;
            else {
                if (bytes) h->bytes_refused = true;
                (void)upload_symbols_narrow<uint16_t>(h, sym, n, staging, (uint16_t *)((char *)staging - narrow_bytes));
                went = 1;
            }
        } else HIP_CHECK(hipMemcpyAsync(staging, sym, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
        h->narrow_upload = went;
        sym = staging;
    } else h->narrow_upload = 0;
    h->stats = Stats();
    h->arena.high = 0;
    Ctx ctx;
    ctx.knobs = kn;
    ctx.stream = h->stream;
    ctx.arena = &h->arena;
    ctx.stats = &h->stats;
    ctx.prof = &h->prof;
    ctx.lean = lean;
    // The build is queued WITHOUT waiting for the device wherever the previous build on this handle says what
    // to expect (alphabet size, no large tie groups): one read-back at the end finds out whether it was
    // right.  If not -- or on a handle's first build -- the build runs with its read-backs in place.
    u32 flags[FLAG_WORDS] = {0};
    auto run = [&](bool spec, bool spec_rounds) -> bool {
        ctx.spec = spec;
        ctx.spec_rounds = spec && spec_rounds;
        ctx.plan_wide = spec ? h->plan_wide : -1;        // (a build that waits for the alphabet plans from its own sample)
        ctx.plan_fused = spec ? h->plan_fused : -1;
        ctx.plan_ht = spec ? h->plan_ht : -1;
        ctx.plan_persist = spec ? h->plan_persist : -1;
        try {
            build_impl(h, ctx, sym, n, (u32)n_docs, doc_offsets, n_strings, h->hint_sigma, tagged);
        } catch (const SpecAbort &) {
            HIP_CHECK(hipStreamSynchronize(h->stream));
            return false;
        } catch (const EastError &) {
            if (!spec) throw;                            // (whatever a wrong guess ran into: the build is repeated without guesses)
            (void)hipStreamSynchronize(h->stream);
            return false;
        }
        HIP_CHECK(hipEventRecord(h->ev1, h->stream));
        HIP_CHECK(hipMemcpyAsync(flags, h->code_map + TEXT_SYMBOLS, sizeof(flags), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(sync_stream(h->stream));
        return true;
    };
    HIP_CHECK(hipEventRecord(h->ev0, h->stream));
    // (the alphabet is guessed whenever the last build took the window sort; that no tie group is large only if it found none)
    const bool speculate = kn.speculate && kn.window_sort && h->hint_valid && h->hint_window && h->hint_sigma <= 254 && !tagged;
    const bool spec_rounds = speculate && h->hint_no_rounds;
    const bool went_through = run(speculate, spec_rounds);
    if (speculate && (!went_through || (flags[FLAG_STATUS] & STATUS_SIGMA_GUESS) ||
                      (spec_rounds && (flags[FLAG_KEEP] || flags[FLAG_FAIL])))) {
        if (g_trace) fprintf(stderr, "[east_hip] speculative build guessed wrong: building again\n");
        h->stats = Stats();
        run(false, false);
    }
    const u32 status = flags[FLAG_STATUS];
    if (status & STATUS_NO_TERMINATOR)
        east_throw(EAST_HIP_ERR_DOMAIN, "a document does not end in a string terminator (>= U+0A00)");
    if (flags[FLAG_CAPPED] && !(status & STATUS_N_STRINGS)) {
        finish_capped_lcp(h, ctx);
        HIP_CHECK(hipEventRecord(h->ev1, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
    }
    if (status & STATUS_N_STRINGS)
        east_throw(EAST_HIP_ERR_DOMAIN, tagged ? "n_strings does not match the tagged terminators found in a document"
                                               : "n_strings does not match the terminators found in a document "
                                                 "(text symbols must be < U+0A00 unless the terminators are tagged)");
    if (flags[FLAG_KG_BAD]) h->kg_marked = false;    // (the score side then builds its k-gram tables itself)
    h->hint_valid = !tagged || h->sigma_hi == 0;
    h->hint_sigma = h->sigma_t;
    h->hint_window = h->stats.window_sorted != 0;
    h->plan_wide = h->stats.window_sorted ? ctx.did_wide : -1;
    // (hardly anything tied behind the wide window -- another kind of text on the same handle: back to the estimate)
    if (ctx.did_wide && h->stats.first_n > 0 && h->stats.first_kept * 50 < h->stats.first_n) h->plan_wide = -1;
    h->plan_fused = h->stats.window_sorted ? ctx.did_fused : -1;
    h->plan_ht = h->stats.window_sorted ? ctx.did_ht : -1;
    h->plan_persist = h->stats.window_sorted ? ctx.did_persist : -1;
    // (the same for the fused finish: it handed more than a few per cent of the suffixes to the rounds, or the separate
    // placement pass left next to nothing -- another kind of text than the plan was made for: the next build decides anew)
    if (h->stats.first_n > 0 && ((ctx.did_fused && h->stats.first_kept * 20 > h->stats.first_n) ||
                                 (!ctx.did_fused && h->stats.first_kept * 50 < h->stats.first_n)))
        h->plan_fused = -1;
    h->hint_no_rounds = h->stats.window_sorted && h->stats.refine_rounds == 0 && !h->stats.long_repeats;
    h->prof.collect();
    HIP_CHECK(hipEventElapsedTime(&h->last_build_ms, h->ev0, h->ev1));
    h->n = n;
    if (h->n_docs != (u32)n_docs) h->n_kp = 0;       // resident keyphrase scratch is sized per n_docs
    h->n_docs = (u32)n_docs;
    h->h_doc_off.assign(doc_offsets, doc_offsets + n_docs + 1);
    h->h_n_strings.assign(n_strings, n_strings + n_docs);
    h->built = true;
    ring_pin_later(h);
}

// ---------------------------------------------------------------- text prep --
#define TP_WORD_HI_WORDS ((0x110000u - TP_TEXT_LIMIT + 31u) / 32u)

// ---- the streamed preparation (textprep.h, "the streamed preparation") --------------------------------------------
// -1: streamed for inputs of TP_STREAM_MIN bytes or more, in about TP_STREAM_CHUNKS chunks; 0: never; > 0: always, in chunks
// of about that many bytes (east_hip_debug_set_text_stream: the tests push the fixtures through chunks of a few dozen bytes)
#define TP_STREAM_MIN ((u32)8 << 20)
#define TP_STREAM_CHUNKS 4

struct TpChunk {
    u32 b0 = 0, b1 = 0;             // bytes [b0, b1) of the concatenated stream (separators included)
    u32 doc_first = 0, n_docs = 0;  // the documents it touches
    bool cont_in = false, cont_out = false;
    std::vector<u32> text_off;      // n_docs + 1: where they start, relative to b0 (the last entry = b1 - b0)
};

// byte p of the concatenated stream (document d holds it)
static inline u32 tp_byte_at(const uint8_t *bytes, const uint8_t *const *texts, const i64 *text_offsets, u32 d, u32 p)
{
    if (p + 1 == (u32)text_offsets[d + 1]) return 0xFFu;              // the separator
    return texts ? texts[d][p - (u32)text_offsets[d]] : bytes[p];
}

// Cuts of the stream where neither a token nor a UTF-8 unit can span them: behind a separator, or behind an ASCII byte
// that is no word character (looked for in the 4 KiB in front of where the chunk would end; a document without one there
// -- one endless token, binary junk -- stays whole).
static std::vector<TpChunk> tp_plan_chunks(const uint8_t *bytes, const uint8_t *const *texts, const i64 *text_offsets, u32 D,
                                           u32 n_bytes, u32 chunk_bytes, const uint8_t *cls256)
{
    std::vector<TpChunk> chunks;
    u32 pos = 0, d = 0;                                  // d: the document that holds byte pos
    while (pos < n_bytes) {
        u32 cut = n_bytes;
        // (the first chunk is a quarter of the others: the preparation -- the slower side -- starts that much earlier)
        const u32 first_div = getenv("EAST_HIP_TP_FIRST_DIV") ? (u32)std::max(1, atoi(getenv("EAST_HIP_TP_FIRST_DIV"))) : 4u;   // (experiments)
        const u32 want = pos == 0 && chunk_bytes >= 4096u ? chunk_bytes / first_div : chunk_bytes;
        if ((u64)pos + want < n_bytes) {
            const u32 target = pos + want;
            u32 dt = d;
            while ((u32)text_offsets[dt + 1] < target) dt++;         // the document that holds byte target - 1
            cut = (u32)text_offsets[dt + 1];                         // (its end, unless a cut inside it is found)
            const u32 lowest = std::max(pos + 1, target > 4096u ? target - 4096u : 0u);
            for (u32 q = target; q-- > lowest;) {
                if (q < (u32)text_offsets[dt]) { cut = (u32)text_offsets[dt]; break; }     // (the document starts in the window: cut in front of it)
                const u32 c = tp_byte_at(bytes, texts, text_offsets, dt, q);
                if (c == 0xFFu || (c < 0x80u && !(cls256[c] & TP_CLASS_WORD))) { cut = q + 1; break; }
            }
        }
        TpChunk ch;
        ch.b0 = pos; ch.b1 = cut;
        while ((u32)text_offsets[d + 1] <= pos) d++;
        ch.doc_first = d;
        ch.cont_in = pos > (u32)text_offsets[d];
        u32 dl = d;
        ch.text_off.push_back(0);
        while ((u32)text_offsets[dl + 1] < cut) { ch.text_off.push_back((u32)text_offsets[dl + 1] - pos); dl++; }
        ch.text_off.push_back(cut - pos);
        ch.n_docs = dl - d + 1;
        ch.cont_out = cut < (u32)text_offsets[dl + 1];
        chunks.push_back(std::move(ch));
        pos = cut;
    }
    return chunks;
}

static thread_local std::chrono::steady_clock::time_point g_tp_call_start;     // (EAST_HIP_TRACE: when build_from_texts was entered)

// ---- many separate texts: through a ring of pinned memory -------------------------------------------------------------
// A copy out of pageable memory is pinned in place by the runtime, copied, unpinned: ~45 us of set-up per call, which a
// 64 MiB text hides and 64 texts of 1 MiB do not (2.75 ms against 1.5 ms; 256 x 1 MiB: 11 ms).  Separate texts of less
// than TP_RING_MAX_TEXT bytes on average therefore go through TP_RING_SLOTS slots of pinned memory: a few host threads
// copy the stream -- text bytes and the 0xFF separators -- into a slot, each its share, while the slots before it are on
// their way to the device (one DMA per slot and chunk, no set-up); the uploader thread alone talks to the runtime.
#define TP_RING_MAX_TEXT ((u64)8 << 20)
#define TP_RING_FIRST_TEXTS 128u              // a handle's first call pins the ring in line only for this many texts or more

// bytes [a, b) of the concatenated stream (texts d with their 0xFF separators, text_offsets as in build_from_texts) -> dst
static void tp_fill_stream(char *dst, u64 a, u64 b, const uint8_t *const *texts, const i64 *text_offsets, u32 D)
{
    u32 d = (u32)(std::upper_bound(text_offsets, text_offsets + D + 1, (i64)a) - text_offsets) - 1u;
    while (a < b) {
        const u64 t0 = (u64)text_offsets[d], sep = (u64)text_offsets[d + 1] - 1u;      // text d = [t0, sep), then its separator
        if (a < sep) {
            const u64 e = std::min(b, sep);
            memcpy(dst, texts[d] + (a - t0), (size_t)(e - a));
            dst += e - a;
            a = e;
        }
        if (a == sep && a < b) { *dst++ = (char)0xFF; a++; }
        if (a > sep) d++;
    }
}

// a first call went without the ring (see prepare_texts_streamed): pin it now that the call is over, in the background
static void ring_pin_later(east_hip_index *h)
{
    ring_adopt(h, false);                  // (a thread that failed to pin is joined here, and the next call may try again)
    if (!h->ring_wanted || h->ring || h->ring_alloc.joinable() || h->ring_pending.load()) return;
    h->ring_wanted = false;
    h->ring_done.store(false);
    const int dev = h->device;
    std::atomic<char *> *slot = &h->ring_pending;
    std::atomic<bool> *done = &h->ring_done;
    h->ring_alloc = std::thread([dev, slot, done]() {
        void *p = nullptr;
        if (hipSetDevice(dev) == hipSuccess && hipHostMalloc(&p, TP_RING_SLOT * TP_RING_SLOTS, hipHostMallocDefault) == hipSuccess) slot->store((char *)p);
        else (void)hipGetLastError();
        done->store(true);
    });
}

// Prepares the collection chunk by chunk; the symbols end up in h->prep_sym, the per-document offsets and string counts in
// h_off / h_m.  Returns false when the monolithic preparation has to take over: kept text at or above U+0A00 (the tagged
// encoding rewrites terminators the chunks no longer remember).  d_bytes: n_bytes + 32 bytes of the arena, nothing uploaded yet.
static bool prepare_texts_streamed(east_hip_index *h, Ctx &ctx, const uint8_t *bytes, const uint8_t *const *texts,
                                   const i64 *text_offsets, u32 D, u32 n_bytes, u32 chunk_bytes, uint8_t *d_bytes,
                                   const TpTables &tables, const uint8_t *d_cls256, const u32 *d_up256,
                                   std::vector<u32> &h_off, std::vector<u32> &h_m)
{
    Arena &ar = *ctx.arena;
    const std::vector<TpChunk> chunks = tp_plan_chunks(bytes, texts, text_offsets, D, n_bytes, chunk_bytes, h->tp_host_tables.data());
    const u32 C = (u32)chunks.size();
    u32 nb_max = 0, dl_max = 0;
    for (const TpChunk &c : chunks) { nb_max = std::max(nb_max, c.b1 - c.b0); dl_max = std::max(dl_max, c.n_docs); }
    if (!h->copy_stream) HIP_CHECK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    while (h->copy_events.size() < C) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->copy_events.push_back(e);
    }
    // the symbols: every kept code point one, every string of three tokens (of three code points or more) a terminator,
    // an empty document two
    const size_t sym_cap = (size_t)n_bytes + (size_t)n_bytes / 9 + 2 * (size_t)D + 64;
    if (sym_cap * 4 > h->prep_cap) {
        HIP_CHECK(hipStreamSynchronize(h->stream));
        if (h->prep_sym) HIP_CHECK(hipFree(h->prep_sym));
        h->prep_sym = nullptr;
        h->prep_cap = 0;
        void *p = nullptr;
        if (hipMalloc(&p, sym_cap * 4) != hipSuccess) east_throw(EAST_HIP_ERR_OOM, "hipMalloc of the prepared symbols failed");
        h->prep_sym = (u32 *)p;
        h->prep_cap = sym_cap * 4;
    }
    // ---- device state shared by the chunks ----
    TpCarry *carry = ar.alloc<TpCarry>(2);
    u32 *d_high = ar.alloc<u32>(1);
    u32 *doc_sym_off_all = ar.alloc<u32>((size_t)D + 1), *m_all = ar.alloc<u32>(D);
    HIP_CHECK(hipMemsetAsync(carry, 0, 2 * sizeof(TpCarry), h->stream));
    HIP_CHECK(hipMemsetAsync(d_high, 0, 4, h->stream));
    // per-chunk scratch (sized for the largest chunk, used by one chunk after the other)
    const u32 ub_tok = nb_max / 2 + 2;                   // a token needs a character and something behind it
    u32 *d_text_off = ar.alloc<u32>((size_t)dl_max + 1);
    u32 *byte_prefix = ar.alloc<u32>((size_t)nb_max / TP_RANK_BLOCK + 2), *tok_prefix = ar.alloc<u32>((size_t)nb_max / TP_RANK_BLOCK + 2);
    // (the counts of the chunks: on the copy stream, each into a stretch of its own -- the host reads chunk c's while
    // chunk c + 1's may already be written)
    std::vector<u32> cnt_off(C + 1, 0);
    for (u32 c = 0; c < C; c++) cnt_off[c + 1] = cnt_off[c] + ceil_div_u32((u64)(chunks[c].b1 - chunks[c].b0) + 1, SCAN_TILE);
    u32 *cp_sums_aux = ar.alloc<u32>(cnt_off[C]);
    std::vector<std::vector<u32>> h_counts(C);
    for (u32 c = 0; c < C; c++) h_counts[c].resize(cnt_off[c + 1] - cnt_off[c]);
    u32 *cpu = ar.alloc<u32>(nb_max);
    uint8_t *cw = ar.alloc<uint8_t>((size_t)nb_max + 32);
    u32 *doc_cp_off = ar.alloc<u32>((size_t)dl_max + 1);
    u32 *tstart = ar.alloc<u32>(ub_tok), *tend = ar.alloc<u32>(ub_tok);
    // (tok_nd, keep and klen side by side: one fill per chunk)
    u32 *tok_nd = ar.alloc<u32>(3 * ((size_t)ub_tok + 1)), *keep = tok_nd + ((size_t)ub_tok + 1), *klen = keep + ((size_t)ub_tok + 1);
    u32 *keep_ex = ar.alloc<u32>((size_t)ub_tok + 1), *klen_ex = ar.alloc<u32>((size_t)ub_tok + 1);
    uint4 *tok_rec = ar.alloc<uint4>(ub_tok);
    u32 *first_tok = ar.alloc<u32>((size_t)dl_max + 1), *n_loc = ar.alloc<u32>((size_t)dl_max + 1);
    u32 *off_loc = ar.alloc<u32>((size_t)dl_max + 1), *kept_tot = ar.alloc<u32>(dl_max), *chars_tot = ar.alloc<u32>(dl_max);

    // ---- the uploads: a thread of their own (a copy out of pageable memory returns when it is staged) ----
    // (ONE uploader: two threads with a copy stream each, the chunks' halves side by side, were measured and are slower --
    // 2.9 against 1.95 ms for 64 MiB: the staging copies of the runtime do not run side by side.  More, smaller chunks
    // towards the end -- a shorter tail behind the last upload -- lose to their launches and read-backs: 2.3 ms with six.)
    std::atomic<int> uploaded{0}, upload_failed{0}, upload_abort{0};
    // The copy stream writes d_bytes (the bottom of the arena) and the chunks' counts: it must not start before what is
    // still queued on the handle's stream -- a score call of the index before, say, reading its tables in the arena --
    // has finished ("one HIP stream per handle": calls are ordered).  ev0 was recorded on h->stream when this call began.
    HIP_CHECK(hipStreamWaitEvent(h->copy_stream, h->ev0, 0));
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    const double t_before = std::chrono::duration<double, std::milli>(t_begin - g_tp_call_start).count();
    std::atomic<double> t_first_fill{0.0}, t_first_dma{0.0};
    std::vector<double> t_up(C, 0.0), t_cnt(C, 0.0), t_queued(C, 0.0);     // (EAST_HIP_TRACE: when a chunk was staged / counted / queued)
    const int device = h->device;
    hipStream_t copy_stream = h->copy_stream;
    const std::vector<hipEvent_t> &events = h->copy_events;
    // (the ring: see tp_fill_stream above)
    // Pinning the ring costs 3-5 ms (hipHostMalloc of 24 MiB), a copy out of pageable memory ~45 us: a handle's FIRST call
    // takes the ring only where that pays at once (TP_RING_FIRST_TEXTS texts or more) -- otherwise it goes the old way and
    // leaves the pinning to a background thread, for the calls after it (`east keyphrases table` over a few dozen files
    // is one call: 64 texts of 1 MiB, first call 11.4-13 ms with the ring pinned in line, second call 5.9).
    ring_adopt(h, ctx.knobs.tp_ring > 0);
    const bool ring_shape = texts && ctx.knobs.tp_ring != 0 && (ctx.knobs.tp_ring > 0 || (D >= 4 && (u64)n_bytes / D < TP_RING_MAX_TEXT));
    const bool use_ring = ring_shape && (h->ring || ctx.knobs.tp_ring > 0 || D >= TP_RING_FIRST_TEXTS);
    if (ring_shape && !use_ring && !h->ring_alloc.joinable() && !h->ring_pending.load()) h->ring_wanted = true;   // (pinned when this call is over: ring_pin_later)
    const size_t ring_slot = ctx.knobs.tp_ring_slot;
    const u32 n_slots = use_ring ? ceil_div_u32(n_bytes, ring_slot) : 0u;
    // (fill threads: three -- measured on the 256-thread host of the MI355X box, 64 texts of 1 MiB: 4 threads 2.25 ms of
    // preparation, 8: 2.3-2.6, 16: 2.6, 32: 2.95 -- starting the threads costs more than their copies save; a 16 MiB
    // chunk is staged in 0.45 ms either way, 37 GB/s)
    static const int ring_threads_env = getenv("EAST_HIP_RING_THREADS") ? atoi(getenv("EAST_HIP_RING_THREADS")) : 0;     // (experiments)
    const int n_fill = !use_ring ? 0 : ring_threads_env > 0 ? std::min(ring_threads_env, 64)
                                     : (int)std::min<u32>(3u, std::max<u32>(2u, std::thread::hardware_concurrency() / 2u));
    if (use_ring && !h->ring) ring_adopt(h, true);       // (a background pin under way: its ring, not a second one)
    if (use_ring && h->ring && h->ring_events.empty())
        for (int i = 0; i < TP_RING_SLOTS; i++) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            h->ring_events.push_back(e);
        }
    if (use_ring && !h->ring) {
        void *p = nullptr;
        if (hipHostMalloc(&p, TP_RING_SLOT * TP_RING_SLOTS, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); east_throw(EAST_HIP_ERR_OOM, "hipHostMalloc of the upload ring failed"); }
        h->ring = (char *)p;
        if (g_trace)
            fprintf(stderr, "[east_hip] text preparation: pinned ring of %zu MiB allocated, %.2f ms into the call\n", (TP_RING_SLOT * TP_RING_SLOTS) >> 20,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_tp_call_start).count());
        for (int i = 0; i < TP_RING_SLOTS; i++) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            h->ring_events.push_back(e);
        }
    }
    std::vector<std::atomic<int>> slot_parts(n_slots);         // fill threads done with their share of a slot
    for (auto &a : slot_parts) a.store(0, std::memory_order_relaxed);
    std::atomic<u32> slots_free{TP_RING_SLOTS};                 // stream slots [0, slots_free) may be filled (their ring slot's last DMA is done)
    char *ring = h->ring;
    std::vector<std::thread> fillers;
    for (int j = 0; j < n_fill; j++)
        fillers.emplace_back([&, j]() {
            for (u32 sl = 0; sl < n_slots; sl++) {
                while (slots_free.load(std::memory_order_acquire) <= sl) {
                    if (upload_abort.load(std::memory_order_acquire)) return;
                    std::this_thread::yield();
                }
                const u64 a = (u64)sl * ring_slot, len = std::min<u64>(ring_slot, (u64)n_bytes - a);
                const u64 lo = a + len * (u64)j / (u64)n_fill, hi = a + len * (u64)(j + 1) / (u64)n_fill;
                if (hi > lo) tp_fill_stream(ring + (size_t)(sl % TP_RING_SLOTS) * ring_slot + (lo - a), lo, hi, texts, text_offsets, D);
                if (slot_parts[sl].fetch_add(1, std::memory_order_release) + 1 == n_fill && sl == 0) t_first_fill.store(since());
            }
        });
    const std::vector<hipEvent_t> &ring_events = h->ring_events;
    std::thread uploader([&, device, copy_stream]() {
        bool ok = hipSetDevice(device) == hipSuccess;
        if (ok && texts && !use_ring) ok = hipMemsetAsync(d_bytes, 0xFF, n_bytes, copy_stream) == hipSuccess;       // the separators
        if (ok) ok = hipMemsetAsync(d_bytes + n_bytes, 0, 32, copy_stream) == hipSuccess;
        u32 sl_next = 0;                                          // (ring) the next stream slot to send, and how far it has been sent
        u64 sent = 0;
        for (u32 c = 0; c < C && ok && !upload_abort.load(std::memory_order_acquire); c++) {
            const TpChunk &ch = chunks[c];
            if (use_ring) {
                // the chunk's bytes: the pieces of the slots it overlaps, one DMA each; a slot is handed back to the fill
                // threads when the DMA of its last piece is done (waited for one slot behind, so that the next is queued)
                while (ok && sent < ch.b1) {
                    while (slot_parts[sl_next].load(std::memory_order_acquire) < n_fill) {
                        if (upload_abort.load(std::memory_order_acquire)) { ok = false; break; }
                        std::this_thread::yield();
                    }
                    if (!ok) break;
                    const u64 s_end = std::min<u64>((u64)(sl_next + 1) * ring_slot, n_bytes), e = std::min<u64>(s_end, ch.b1);
                    ok = hipMemcpyAsync(d_bytes + sent, ring + (size_t)(sl_next % TP_RING_SLOTS) * ring_slot + (sent - (u64)sl_next * ring_slot),
                                        (size_t)(e - sent), hipMemcpyHostToDevice, copy_stream) == hipSuccess;
                    if (sent == 0) t_first_dma.store(since());
                    sent = e;
                    if (ok && sent == s_end) {
                        ok = hipEventRecord(ring_events[sl_next % TP_RING_SLOTS], copy_stream) == hipSuccess;
                        if (ok && sl_next >= 1) {
                            ok = hipEventSynchronize(ring_events[(sl_next - 1) % TP_RING_SLOTS]) == hipSuccess;
                            slots_free.store(sl_next - 1 + 1 + TP_RING_SLOTS, std::memory_order_release);
                        }
                        sl_next++;
                    }
                }
            } else if (texts) {
                for (u32 i = 0; i < ch.n_docs && ok; i++) {
                    const u32 d = ch.doc_first + i;
                    const u32 lo = std::max(ch.b0, (u32)text_offsets[d]), hi = std::min(ch.b1, (u32)text_offsets[d + 1] - 1u);    // (without the separator)
                    if (hi > lo)
                        ok = hipMemcpyAsync(d_bytes + lo, texts[d] + (lo - (u32)text_offsets[d]), hi - lo, hipMemcpyHostToDevice,
                                            copy_stream) == hipSuccess;
                }
            } else {
                ok = hipMemcpyAsync(d_bytes + ch.b0, bytes + ch.b0, ch.b1 - ch.b0, hipMemcpyHostToDevice, copy_stream) == hipSuccess;
            }
            if (ok) {
                // the chunk's code point count (per tile of the scan; the host adds them up), behind its bytes
                const u32 nb = ch.b1 - ch.b0, nb_cp = ceil_div_u32((u64)nb + 1, SCAN_TILE);
                hipLaunchKernelGGL((scan_reduce_kernel<TpStartIn>), dim3(nb_cp), dim3(BLOCK), 0, copy_stream,
                                   TpStartIn{d_bytes + ch.b0, nb}, nb + 1, cp_sums_aux + cnt_off[c]);
                ok = hipGetLastError() == hipSuccess &&
                     hipMemcpyAsync(h_counts[c].data(), cp_sums_aux + cnt_off[c], (size_t)nb_cp * 4, hipMemcpyDeviceToHost,
                                    copy_stream) == hipSuccess;
            }
            if (ok) ok = hipEventRecord(events[c], copy_stream) == hipSuccess;
            t_up[c] = since();
            if (ok) uploaded.store((int)c + 1, std::memory_order_release);
        }
        if (!ok) { (void)hipGetLastError(); upload_failed.store(1, std::memory_order_release); }
    });
    // (unwinding -- a HIP error or a thrown status on the compute side: the uploader stops queueing, and nothing it has
    // queued may still be writing the arena or the host-side counts when they are released)
    struct Joiner {
        std::thread &t;
        std::vector<std::thread> &fill;
        std::atomic<int> &abort;
        hipStream_t copy;
        ~Joiner()
        {
            if (t.joinable()) {                          // (the regular path has joined already)
                abort.store(1, std::memory_order_release);
                t.join();
                (void)hipStreamSynchronize(copy);
            }
            for (auto &f : fill)
                if (f.joinable()) f.join();
        }
    } joiner{uploader, fillers, upload_abort, copy_stream};

    for (u32 c = 0; c < C; c++) {
        const TpChunk &ch = chunks[c];
        const u32 nb = ch.b1 - ch.b0, Dl = ch.n_docs;
        const uint8_t *b = d_bytes + ch.b0;
        // the chunk's bytes: recorded by the uploader, waited for by the compute stream
        while (uploaded.load(std::memory_order_acquire) <= (int)c) {
            if (upload_failed.load(std::memory_order_acquire)) east_throw(EAST_HIP_ERR_HIP, "upload of the raw text failed");
            std::this_thread::yield();
        }
        HIP_CHECK(hipStreamWaitEvent(h->stream, events[c], 0));
        HIP_CHECK(hipEventSynchronize(events[c]));         // (the host reads the chunk's counts)
        t_cnt[c] = since();
        HIP_CHECK(hipMemcpyAsync(d_text_off, ch.text_off.data(), ((size_t)Dl + 1) * 4, hipMemcpyHostToDevice, h->stream));
        // bytes -> code points (the count first: a chunk in which every byte is a code point of its own needs no index).
        // The count only needs the chunk's bytes: the uploader queues it on the copy stream right behind them (and records
        // the event behind it), so that the host has the answer -- and queues the chunk's kernels -- while the chunk
        // before is still being prepared.
        const u32 nb_cp = ceil_div_u32((u64)nb + 1, SCAN_TILE);
        u32 n_cp = 0;
        for (u32 i = 0; i < nb_cp; i++) n_cp += h_counts[c][i];
        const bool bytewise = n_cp == nb;
        if (bytewise) {
            LAUNCH(ctx, tp_classify_bytes_kernel, ceil_div_u32(nb, BLOCK * 16), b, nb, d_cls256, cw);
        } else {
            const u32 n_bblk = ceil_div_u32(nb, TP_RANK_BLOCK);
            LAUNCH(ctx, (tp_block_counts_kernel<TpStartIn>), ceil_div_u32((u64)n_bblk + 1, 8), TpStartIn{b, nb}, nb, n_bblk, byte_prefix);
            device_scan<ArrIn, false>(ctx, ArrIn{byte_prefix}, n_bblk + 1, byte_prefix);
            LAUNCH(ctx, tp_decode_kernel, ceil_div_u32(nb, BLOCK), b, nb, (const u32 *)byte_prefix, tables, cpu, cw);
        }
        LAUNCH(ctx, tp_doc_cp_offsets_kernel, ceil_div_u32(Dl + 1, WAVES_PER_BLOCK), b, nb, bytewise ? (const u32 *)nullptr : (const u32 *)byte_prefix,
               (const u32 *)d_text_off, Dl, doc_cp_off);
        // code points -> tokens (their number stays on the device: the last entry of the blocks' prefix sums)
        const u32 n_tblk = ceil_div_u32(n_cp, TP_RANK_BLOCK);
        LAUNCH(ctx, (tp_block_counts_kernel<TpTokStartIn>), ceil_div_u32((u64)n_tblk + 1, 8), TpTokStartIn{cw, n_cp}, n_cp, n_tblk, tok_prefix);
        device_scan<ArrIn, false>(ctx, ArrIn{tok_prefix}, n_tblk + 1, tok_prefix);
        const u32 *n_tok_dev = tok_prefix + n_tblk;
        const u32 ub = n_cp / 2 + 2;
        // (everything over the tokens is bounded by their number on the device -- a third to a quarter of the upper bound ub:
        // the zeroing, and ONE scan for kept tokens and kept symbols together, scan.h: device_scan_pair_bounded)
        LAUNCH(ctx, tp_zero_tokens_kernel, ceil_div_u32((u64)ub + 1, BLOCK * 4), tok_nd, keep, klen, ub, n_tok_dev);
        LAUNCH(ctx, tp_token_bounds_kernel, ceil_div_u32(n_cp, BLOCK * TP_VEC), (const uint8_t *)cw, (const u32 *)tok_prefix, n_cp, tstart,
               tend, tok_nd);
        LAUNCH(ctx, tp_token_keep_kernel, ceil_div_u32(ub, BLOCK), (const u32 *)tstart, (const u32 *)tend, (const u32 *)tok_nd, ub, keep,
               klen, n_tok_dev);
        device_scan_pair_bounded(ctx, keep, klen, ub + 1, n_tok_dev, 1u, keep_ex, klen_ex);
        // tokens -> the documents' strings and symbols, with what earlier chunks emitted of the first document
        const TpCarry *cin = carry + (c & 1u);
        TpCarry *cout = carry + ((c + 1u) & 1u);
        LAUNCH(ctx, tp_stream_docs_kernel, ceil_div_u32(Dl + 1, WAVES_PER_BLOCK), (const u32 *)doc_cp_off, (const uint8_t *)cw, n_cp, (const u32 *)tok_prefix, (const u32 *)keep_ex,
               (const u32 *)klen_ex, Dl, (u32)ch.cont_in, (u32)ch.cont_out, cin, first_tok, n_loc, kept_tot, chars_tot);
        device_scan<ArrIn, false>(ctx, ArrIn{n_loc}, Dl + 1, off_loc);
        LAUNCH(ctx, tp_stream_token_out_kernel, ceil_div_u32(ub, BLOCK), (const u32 *)tstart, (const u32 *)tend, (const u32 *)keep_ex,
               (const u32 *)klen_ex, (const u32 *)doc_cp_off, (const u32 *)first_tok, (const u32 *)off_loc, (const u32 *)kept_tot, Dl,
               (u32)ch.cont_in, (u32)ch.cont_out, cin, n_tok_dev, tok_rec);
        LAUNCH(ctx, tp_emit_kernel, ceil_div_u32(n_cp, BLOCK), bytewise ? (const u32 *)nullptr : (const u32 *)cpu, b, d_up256,
               (const uint8_t *)cw, (const u32 *)tok_prefix, (const uint4 *)tok_rec, n_cp, h->prep_sym, d_high);
        LAUNCH(ctx, tp_stream_close_docs_kernel, ceil_div_u32(Dl, BLOCK), (const u32 *)off_loc, (const u32 *)n_loc, (const u32 *)kept_tot,
               (const u32 *)chars_tot, Dl, ch.doc_first, (u32)ch.cont_in, (u32)ch.cont_out, cin, cout, doc_sym_off_all, m_all,
               h->prep_sym);
        t_queued[c] = since();
    }
    uploader.join();
    for (auto &f : fillers) f.join();
    // the total, the per-document offsets and string counts, "kept text at or above U+0A00"
    h_off.resize((size_t)D + 1);
    h_m.resize(D);
    u32 high = 0;
    TpCarry last;
    HIP_CHECK(hipEventRecord(h->ev1, h->stream));
    HIP_CHECK(hipMemcpyAsync(h_off.data(), doc_sym_off_all, (size_t)D * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipMemcpyAsync(h_m.data(), m_all, (size_t)D * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipMemcpyAsync(&last, carry + (C & 1u), sizeof(last), hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipMemcpyAsync(&high, d_high, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    if (g_trace) {
        fprintf(stderr, "[east_hip] streamed preparation, %u chunks (ms since its start: staged / counted / queued):", C);
        for (u32 c = 0; c < C; c++) fprintf(stderr, " [%u MiB %.2f %.2f %.2f]", (chunks[c].b1 - chunks[c].b0) >> 20, t_up[c], t_cnt[c], t_queued[c]);
        fprintf(stderr, " done %.2f", since());
        if (use_ring) fprintf(stderr, "; %.2f ms of the call in front of it, first ring slot filled at %.2f, its DMA queued at %.2f", t_before,
                              t_first_fill.load(), t_first_dma.load());
        fprintf(stderr, "\n");
    }
    h_off[D] = last.sym_base;
    return high == 0;
}

// bytes: the texts concatenated, each followed by one 0xFF byte (host pointer).
// (texts != nullptr: the texts lie apart in host memory -- text d = texts[d], text_offsets as if they were
// concatenated with their separators; they are uploaded one by one and never joined on the host)
static void build_from_texts(east_hip_index *h, const uint8_t *bytes, i64 n_bytes64, const i64 *text_offsets,
                             int32_t n_docs, const uint8_t *cp_class, const u32 *cp_upper, const u32 *word_hi,
                             const u32 *digit_hi, const u32 *hi_upper_from, const u32 *hi_upper_to, int32_t n_hi_upper,
                             const uint8_t *const *texts = nullptr)
{
    g_tp_call_start = std::chrono::steady_clock::now();
    if (!h) east_throw(EAST_HIP_ERR_INVALID, "null handle");
    if ((!bytes && !texts) || !text_offsets || !cp_class || !cp_upper || !word_hi || !digit_hi || n_docs < 1 || n_hi_upper < 0 ||
        (n_hi_upper > 0 && (!hi_upper_from || !hi_upper_to)))
        east_throw(EAST_HIP_ERR_INVALID, "null argument or no documents");
    if (n_bytes64 < n_docs || n_bytes64 >= (i64)0x7FFFFFF0) east_throw(EAST_HIP_ERR_INVALID, "total bytes out of range");
    if (text_offsets[0] != 0 || text_offsets[n_docs] != n_bytes64)
        east_throw(EAST_HIP_ERR_INVALID, "text_offsets must start at 0 and end at the total");
    for (int32_t d = 0; d < n_docs; d++) {
        if (text_offsets[d + 1] <= text_offsets[d]) east_throw(EAST_HIP_ERR_INVALID, "text_offsets must increase");
        if (texts ? (text_offsets[d + 1] - text_offsets[d] > 1 && !texts[d]) : bytes[text_offsets[d + 1] - 1] != 0xFFu)
            east_throw(EAST_HIP_ERR_INVALID, texts ? "null text" : "every text must be followed by one 0xFF separator byte");
    }
    use_device(h);
    h->built = false;
    const u32 n_bytes = (u32)n_bytes64, D = (u32)n_docs;
    const size_t arena_before = h->arena.cap;
    size_t arena_need = (size_t)n_bytes * 46 + (size_t)D * 96 + (8u << 20);
    if (arena_before < arena_need) {
        // (the arena has to grow anyway -- a handle's first call: sized for the build behind the preparation at once, on the
        // most symbols these bytes can turn into, instead of a second hipMalloc + hipFree of gigabytes in the same call)
        const u64 n_upper = std::min<u64>((u64)n_bytes + (u64)n_bytes / 9 + 2 * (u64)D + 64, 0x7FFFFFE0ull);
        size_t free_b = 0, total_b = 0;
        const size_t both = plan_arena_bytes((u32)n_upper, D);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && both < (size_t)(0.5 * (double)(free_b + arena_before))) arena_need = std::max(arena_need, both);
        else (void)hipGetLastError();
    }
    ensure_arena(h, arena_need);
    if (g_trace && h->arena.cap != arena_before)
        fprintf(stderr, "[east_hip] text preparation: arena of %.2f GiB allocated, %.2f ms into the call\n", h->arena.cap / 1073741824.0,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_tp_call_start).count());
    Arena &ar = h->arena;
    ar.release(0);
    ar.high = 0;
    Ctx ctx;
    ctx.stream = h->stream;
    ctx.arena = &ar;
    ctx.stats = &h->stats;
    ctx.prof = &h->prof;
    HIP_CHECK(hipEventRecord(h->ev0, h->stream));

    uint8_t *d_bytes = ar.alloc<uint8_t>((size_t)n_bytes + 32);    // (padding: the byte-class pass loads whole 16-byte groups)
    u32 *d_text_off = ar.alloc<u32>((size_t)D + 1);
    u32 *d_high = ar.alloc<u32>(1);
    std::vector<u32> off32((size_t)D + 1);
    for (u32 d = 0; d <= D; d++) off32[d] = (u32)text_offsets[d];
    // (large inputs: the text goes up chunk by chunk and is prepared as it arrives, see prepare_texts_streamed)
    const u32 stream_chunk = ctx.knobs.tp_stream > 0 ? (u32)std::min<i64>(ctx.knobs.tp_stream, 0x40000000)
                             : ctx.knobs.tp_stream < 0 && n_bytes >= TP_STREAM_MIN ? std::max<u32>(n_bytes / TP_STREAM_CHUNKS + 1, 1u << 20) : 0u;
    auto upload_all = [&]() {
        if (texts) {
            HIP_CHECK(hipMemsetAsync(d_bytes, 0xFF, n_bytes, h->stream));              // the separators
            for (u32 d = 0; d < D; d++) {
                const size_t len = (size_t)(text_offsets[d + 1] - text_offsets[d] - 1);
                if (len) HIP_CHECK(hipMemcpyAsync(d_bytes + text_offsets[d], texts[d], len, hipMemcpyHostToDevice, h->stream));
            }
        } else {
            HIP_CHECK(hipMemcpyAsync(d_bytes, bytes, n_bytes, hipMemcpyHostToDevice, h->stream));
        }
        HIP_CHECK(hipMemsetAsync(d_bytes + n_bytes, 0, 32, h->stream));
    };
    if (!stream_chunk) upload_all();
    HIP_CHECK(hipMemcpyAsync(d_text_off, off32.data(), off32.size() * 4, hipMemcpyHostToDevice, h->stream));
    // The caller's Unicode tables (290 KB) stay on the device between calls (own allocation): they are uploaded again only
    // when their content changes -- a 64-bit hash over all of them, taken while the text is on its way.  With them go the two
    // 256-entry tables of the byte-wise fast path (class and upper-cased code point of a byte that is a code point of its own).
    const size_t tb_class = 0, tb_upper = tb_class + TP_TEXT_LIMIT, tb_word = tb_upper + (size_t)TP_TEXT_LIMIT * 4,
                 tb_digit = tb_word + (size_t)TP_WORD_HI_WORDS * 4, tb_from = tb_digit + (size_t)TP_WORD_HI_WORDS * 4,
                 tb_to = tb_from + ((size_t)n_hi_upper + 1) * 4, tb_cls256 = tb_to + ((size_t)n_hi_upper + 1) * 4,
                 tb_up256 = tb_cls256 + 256, tb_total = tb_up256 + 1024;
    {
        u64 hash = 0x9E3779B97F4A7C15ull ^ (u64)n_hi_upper;
        auto mix = [&](const void *p, size_t bytes) {
            const u64 *q = (const u64 *)p;
            for (size_t i = 0; i < bytes / 8; i++) hash = (hash ^ q[i]) * 0x100000001B3ull + (hash >> 29);
        };
        mix(cp_class, TP_TEXT_LIMIT); mix(cp_upper, (size_t)TP_TEXT_LIMIT * 4); mix(word_hi, (size_t)TP_WORD_HI_WORDS * 4);
        mix(digit_hi, (size_t)TP_WORD_HI_WORDS * 4);
        for (int32_t q = 0; q < n_hi_upper; q++) hash = (hash ^ (((u64)hi_upper_from[q] << 32) | hi_upper_to[q])) * 0x100000001B3ull + (hash >> 29);
        if (!h->tp_tables || h->tp_tables_bytes < tb_total || h->tp_tables_hash != hash) {
            if (h->tp_tables_bytes < tb_total) {
                HIP_CHECK(hipStreamSynchronize(h->stream));
                if (h->tp_tables) HIP_CHECK(hipFree(h->tp_tables));
                h->tp_tables = nullptr;
                h->tp_tables_bytes = 0;
                void *p = nullptr;
                if (hipMalloc(&p, tb_total) != hipSuccess) east_throw(EAST_HIP_ERR_OOM, "hipMalloc of the Unicode tables failed");
                h->tp_tables = (char *)p;
                h->tp_tables_bytes = tb_total;
            }
            h->tp_host_tables.resize(256 + 1024);
            uint8_t *cls256 = h->tp_host_tables.data();
            u32 *up256 = reinterpret_cast<u32 *>(h->tp_host_tables.data() + 256);
            for (u32 x = 0; x < 256; x++) {              // (as tp_decode_kernel: upper first, then the class of the result)
                u32 cp = x < 0x80u ? cp_upper[x] : TP_REPLACEMENT;
                if (cp >= TP_TEXT_LIMIT) {
                    for (int32_t q = 0; q < n_hi_upper; q++)
                        if (hi_upper_from[q] == cp) { cp = hi_upper_to[q]; break; }
                }
                u32 cls;
                if (cp < TP_TEXT_LIMIT) cls = cp_class[cp];
                else { const u32 k = cp - TP_TEXT_LIMIT; cls = ((word_hi[k >> 5] >> (k & 31u)) & 1u) | (((digit_hi[k >> 5] >> (k & 31u)) & 1u) << 1); }
                cls256[x] = (uint8_t)cls;
                up256[x] = cp;
            }
            char *t = h->tp_tables;
            HIP_CHECK(hipMemcpyAsync(t + tb_class, cp_class, TP_TEXT_LIMIT, hipMemcpyHostToDevice, h->stream));
            HIP_CHECK(hipMemcpyAsync(t + tb_upper, cp_upper, (size_t)TP_TEXT_LIMIT * 4, hipMemcpyHostToDevice, h->stream));
            HIP_CHECK(hipMemcpyAsync(t + tb_word, word_hi, (size_t)TP_WORD_HI_WORDS * 4, hipMemcpyHostToDevice, h->stream));
            HIP_CHECK(hipMemcpyAsync(t + tb_digit, digit_hi, (size_t)TP_WORD_HI_WORDS * 4, hipMemcpyHostToDevice, h->stream));
            if (n_hi_upper) {
                HIP_CHECK(hipMemcpyAsync(t + tb_from, hi_upper_from, (size_t)n_hi_upper * 4, hipMemcpyHostToDevice, h->stream));
                HIP_CHECK(hipMemcpyAsync(t + tb_to, hi_upper_to, (size_t)n_hi_upper * 4, hipMemcpyHostToDevice, h->stream));
            }
            HIP_CHECK(hipMemcpyAsync(t + tb_cls256, cls256, 256, hipMemcpyHostToDevice, h->stream));
            HIP_CHECK(hipMemcpyAsync(t + tb_up256, up256, 1024, hipMemcpyHostToDevice, h->stream));
            h->tp_tables_hash = hash;                   // (the read-back below waits for the stream: the host buffers are the caller's / the handle's)
        }
    }
    const uint8_t *d_class = (const uint8_t *)(h->tp_tables + tb_class), *d_cls256 = (const uint8_t *)(h->tp_tables + tb_cls256);
    const u32 *d_upper = (const u32 *)(h->tp_tables + tb_upper), *d_word_hi = (const u32 *)(h->tp_tables + tb_word),
              *d_digit_hi = (const u32 *)(h->tp_tables + tb_digit), *d_hi_from = (const u32 *)(h->tp_tables + tb_from),
              *d_hi_to = (const u32 *)(h->tp_tables + tb_to), *d_up256 = (const u32 *)(h->tp_tables + tb_up256);
    HIP_CHECK(hipMemsetAsync(d_high, 0, 4, h->stream));
    const TpTables tables{d_class, d_upper, d_word_hi, d_digit_hi, d_hi_from, d_hi_to, (u32)n_hi_upper};
    if (stream_chunk) {
        std::vector<u32> s_off, s_m;
        const size_t mark = ar.mark();
        const bool done = prepare_texts_streamed(h, ctx, bytes, texts, text_offsets, D, n_bytes, stream_chunk, d_bytes, tables,
                                                 d_cls256, d_up256, s_off, s_m);
        ar.release(mark);
        if (done) {
            HIP_CHECK(hipEventElapsedTime(&h->last_prep_ms, h->ev0, h->ev1));
            h->prep_tagged = false;
            h->prep_n = s_off[D];
            h->prep_doc_off.resize((size_t)D + 1);
            h->prep_n_strings.resize(D);
            for (u32 d = 0; d <= D; d++) h->prep_doc_off[d] = s_off[d];
            for (u32 d = 0; d < D; d++) h->prep_n_strings[d] = (int32_t)s_m[d];
            build_common(h, h->prep_sym, false, s_off[D], h->prep_doc_off.data(), h->prep_n_strings.data(), n_docs, false);
            ring_pin_later(h);
            return;
        }
        upload_all();                                   // (kept text at or above U+0A00: the preparation in one piece, tagged encoding)
    }

    // bytes -> code points
    // (first only the count: text in which every byte is a code point of its own -- ASCII, Latin-1 junk -- needs no
    // index at all, and the count has to come back to the host anyway)
    // (unit starts in front of every block of 256 bytes -- textprep.h, "ranks without a per-element index" --; the last
    // entry is their total)
    const u32 n_bblk = ceil_div_u32(n_bytes, TP_RANK_BLOCK);
    u32 *byte_prefix = ar.alloc<u32>((size_t)n_bblk + 1);
    LAUNCH(ctx, (tp_block_counts_kernel<TpStartIn>), ceil_div_u32((u64)n_bblk + 1, 8), TpStartIn{d_bytes, n_bytes}, n_bytes, n_bblk,
           byte_prefix);
    device_scan<ArrIn, false>(ctx, ArrIn{byte_prefix}, n_bblk + 1, byte_prefix);
    u32 n_cp = 0;
    HIP_CHECK(hipMemcpyAsync(&n_cp, byte_prefix + n_bblk, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));          // also covers off32
    // (every byte a code point of its own: no code point array -- classes from a byte table, the kept bytes are mapped when
    // they are emitted)
    const bool bytewise = n_cp == n_bytes;
    u32 *cpu = bytewise ? nullptr : ar.alloc<u32>(n_cp);
    uint8_t *cw = ar.alloc<uint8_t>((size_t)n_cp + 32);
    u32 *doc_cp_off = ar.alloc<u32>((size_t)D + 1);
    if (bytewise) {
        LAUNCH(ctx, tp_classify_bytes_kernel, ceil_div_u32(n_bytes, BLOCK * 16), (const uint8_t *)d_bytes, n_bytes,
               (const uint8_t *)d_cls256, cw);
    } else {
        LAUNCH(ctx, tp_decode_kernel, ceil_div_u32(n_bytes, BLOCK), (const uint8_t *)d_bytes, n_bytes, (const u32 *)byte_prefix,
               tables, cpu, cw);
    }
    LAUNCH(ctx, tp_doc_cp_offsets_kernel, ceil_div_u32(D + 1, WAVES_PER_BLOCK), (const uint8_t *)d_bytes, n_bytes,
           bytewise ? (const u32 *)nullptr : (const u32 *)byte_prefix, (const u32 *)d_text_off, D, doc_cp_off);

    // code points -> tokens (token starts in front of every block of 256 code points; the last entry: their number)
    const u32 n_tblk = ceil_div_u32(n_cp, TP_RANK_BLOCK);
    u32 *tok_prefix = ar.alloc<u32>((size_t)n_tblk + 1);
    LAUNCH(ctx, (tp_block_counts_kernel<TpTokStartIn>), ceil_div_u32((u64)n_tblk + 1, 8), TpTokStartIn{cw, n_cp}, n_cp, n_tblk, tok_prefix);
    device_scan<ArrIn, false>(ctx, ArrIn{tok_prefix}, n_tblk + 1, tok_prefix);
    u32 n_tok = 0;
    u32 high = 0;
    HIP_CHECK(hipMemcpyAsync(&n_tok, tok_prefix + n_tblk, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    u32 *tstart = ar.alloc<u32>((size_t)n_tok + 1), *tend = ar.alloc<u32>((size_t)n_tok + 1);
    u32 *keep = ar.alloc<u32>((size_t)n_tok + 1), *klen = ar.alloc<u32>((size_t)n_tok + 1);
    u32 *keep_ex = ar.alloc<u32>((size_t)n_tok + 1), *klen_ex = ar.alloc<u32>((size_t)n_tok + 1);
    u32 *tok_nd = ar.alloc<u32>((size_t)n_tok + 1);      // token holds a character that is not a digit
    HIP_CHECK(hipMemsetAsync(tok_nd, 0, ((size_t)n_tok + 1) * 4, h->stream));
    HIP_CHECK(hipMemsetAsync(keep + n_tok, 0, 4, h->stream));
    HIP_CHECK(hipMemsetAsync(klen + n_tok, 0, 4, h->stream));
    if (n_tok) {
        LAUNCH(ctx, tp_token_bounds_kernel, ceil_div_u32(n_cp, BLOCK * TP_VEC), (const uint8_t *)cw, (const u32 *)tok_prefix, n_cp,
               tstart, tend, tok_nd);
        LAUNCH(ctx, tp_token_keep_kernel, ceil_div_u32(n_tok, BLOCK), (const u32 *)tstart, (const u32 *)tend,
               (const u32 *)tok_nd, n_tok, keep, klen);
    }
    device_scan<ArrIn, false>(ctx, ArrIn{keep}, n_tok + 1, keep_ex);
    device_scan<ArrIn, false>(ctx, ArrIn{klen}, n_tok + 1, klen_ex);

    // tokens -> per-document strings and symbols
    u32 *first_tok = ar.alloc<u32>((size_t)D + 1), *m_d = ar.alloc<u32>(D), *n_d = ar.alloc<u32>((size_t)D + 1);
    u32 *doc_sym_off = ar.alloc<u32>((size_t)D + 1);
    HIP_CHECK(hipMemsetAsync(n_d + D, 0, 4, h->stream));
    LAUNCH(ctx, tp_doc_counts_kernel, ceil_div_u32(D + 1, WAVES_PER_BLOCK), (const u32 *)doc_cp_off, (const uint8_t *)cw, n_cp, (const u32 *)tok_prefix,
           (const u32 *)keep_ex, (const u32 *)klen_ex, D, first_tok, m_d, n_d);
    device_scan<ArrIn, false>(ctx, ArrIn{n_d}, D + 1, doc_sym_off);
    std::vector<u32> h_off((size_t)D + 1), h_m(D);
    HIP_CHECK(hipMemcpyAsync(h_off.data(), doc_sym_off, h_off.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipMemcpyAsync(h_m.data(), m_d, h_m.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    const u32 n_sym = h_off[D];
    if ((size_t)n_sym * 4 > h->prep_cap) {
        if (h->prep_sym) HIP_CHECK(hipFree(h->prep_sym));
        h->prep_sym = nullptr;
        h->prep_cap = 0;
        void *p = nullptr;
        if (hipMalloc(&p, (size_t)n_sym * 4) != hipSuccess) east_throw(EAST_HIP_ERR_OOM, "hipMalloc of the prepared symbols failed");
        h->prep_sym = (u32 *)p;
        h->prep_cap = (size_t)n_sym * 4;
    }
    if (n_tok) {
        u32 *tok_out = keep, *tok_term = klen;           // (keep / klen are dead once their scans exist)
        uint4 *tok_rec = ar.alloc<uint4>(n_tok);
        LAUNCH(ctx, tp_token_out_kernel, ceil_div_u32(n_tok, BLOCK), (const u32 *)tstart, (const u32 *)keep_ex,
               (const u32 *)klen_ex, (const u32 *)doc_cp_off, (const u32 *)first_tok, (const u32 *)doc_sym_off, D, n_tok,
               (const u32 *)tend, tok_out, tok_term, tok_rec);
        LAUNCH(ctx, tp_emit_kernel, ceil_div_u32(n_cp, BLOCK), (const u32 *)cpu, (const uint8_t *)d_bytes,
               (const u32 *)d_up256, (const uint8_t *)cw, (const u32 *)tok_prefix, (const uint4 *)tok_rec, n_cp, h->prep_sym, d_high);
    }
    LAUNCH(ctx, tp_empty_docs_kernel, ceil_div_u32(D, BLOCK), (const u32 *)first_tok, (const u32 *)keep_ex,
           (const u32 *)doc_sym_off, D, h->prep_sym);
    HIP_CHECK(hipEventRecord(h->ev1, h->stream));
    HIP_CHECK(hipMemcpyAsync(&high, d_high, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    HIP_CHECK(hipEventElapsedTime(&h->last_prep_ms, h->ev0, h->ev1));
    h->prep_tagged = high != 0;
    if (h->prep_tagged) {
        // kept word characters at or above U+0A00: the symbols go on in the tagged encoding
        if (n_tok)
            LAUNCH(ctx, tp_tag_terminators_kernel, ceil_div_u32(n_tok, BLOCK), (const u32 *)tstart, (const u32 *)tend,
                   (const u32 *)keep, (const u32 *)klen, n_tok, h->prep_sym);
        LAUNCH(ctx, tp_tag_empty_docs_kernel, ceil_div_u32(D, BLOCK), (const u32 *)first_tok, (const u32 *)keep_ex,
               (const u32 *)doc_sym_off, D, h->prep_sym);
    }

    h->prep_n = n_sym;
    h->prep_doc_off.resize((size_t)D + 1);
    h->prep_n_strings.resize(D);
    for (u32 d = 0; d <= D; d++) h->prep_doc_off[d] = h_off[d];
    for (u32 d = 0; d < D; d++) h->prep_n_strings[d] = (int32_t)h_m[d];
    build_common(h, h->prep_sym, false, n_sym, h->prep_doc_off.data(), h->prep_n_strings.data(), n_docs, h->prep_tagged);
}

// ------------------------------------------------------------------ score --
// The walk writes one fp64 per (keyphrase suffix, document); that scratch is bounded -- a table over a million
// one-line documents would need hundreds of GB -- and the documents are scored a stretch at a time.
static u32 score_doc_chunk(u32 n_q, u32 n_docs, size_t scratch_bytes)
{
    const size_t per_doc = (size_t)n_q * 8;
    const size_t fit = per_doc ? scratch_bytes / per_doc : n_docs;
    return (u32)std::min<size_t>(n_docs, std::max<size_t>(fit, 1));
}

static void set_keyphrases(east_hip_index *h, const u32 *q_symbols, const i64 *q_offsets, int32_t n_kp)
{
    if (!h) east_throw(EAST_HIP_ERR_INVALID, "null handle");
    if (!h->built) east_throw(EAST_HIP_ERR_NOT_BUILT, "no index has been built on this handle");
    if (n_kp < 1 || !q_symbols || !q_offsets) east_throw(EAST_HIP_ERR_INVALID, "no keyphrases");
    if (q_offsets[0] != 0) east_throw(EAST_HIP_ERR_INVALID, "q_offsets[0] must be 0");
    for (int32_t k = 0; k < n_kp; k++)
        if (q_offsets[k + 1] <= q_offsets[k])
            east_throw(EAST_HIP_ERR_INVALID, "empty keyphrase (the reference raises ZeroDivisionError, easa.py:134)");
    const i64 S = q_offsets[n_kp];
    if (S >= (i64)0x7FFFFFF0 || (i64)n_kp * h->n_docs >= ((i64)1 << 40))
        east_throw(EAST_HIP_ERR_INVALID, "keyphrase set too large");
    use_device(h);
    const u32 n_q = (u32)S;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const Knobs kn = knobs_snapshot();
    const u32 chunk = score_doc_chunk(n_q, h->n_docs, kn.score_scratch_bytes);
    const size_t bytes = 256 + al((size_t)n_q * 4) * 3 + al(((size_t)n_kp + 1) * 4) * 3 +
                         al((size_t)n_q * chunk * 8) + al((size_t)n_kp * h->n_docs * 8) * 2;
    if (bytes > h->q_cap) {
        HIP_CHECK(hipStreamSynchronize(h->stream));
        if (h->q_buf) HIP_CHECK(hipFree(h->q_buf));
        h->q_buf = nullptr;
        h->q_cap = 0;
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) east_throw(EAST_HIP_ERR_OOM, "hipMalloc of the score scratch failed");
        h->q_buf = (char *)p;
        h->q_cap = bytes;
    }
    char *p = h->q_buf + 256;                            // (the first bytes hold the probe counter of east_hip_score_probes)
    h->q_raw = (u32 *)p;  p += al((size_t)n_q * 4);
    h->q_code = (u32 *)p; p += al((size_t)n_q * 4);
    h->q_end = (u32 *)p;  p += al((size_t)n_q * 4);
    h->q_off = (u32 *)p;  p += al(((size_t)n_kp + 1) * 4);
    h->group_off = (u32 *)p; p += al(((size_t)n_kp + 1) * 4);      // synonym-expanded scoring: variants per keyphrase
    h->q_blk = (u32 *)p;  p += al(((size_t)n_kp + 1) * 4);
    h->suffix = (double *)p; p += al((size_t)n_q * chunk * 8);
    h->table = (double *)p; p += al((size_t)n_kp * h->n_docs * 8);
    h->table_g = (double *)p;
    std::vector<u32> end(n_q), off((size_t)n_kp + 1);
    for (int32_t k = 0; k < n_kp; k++) {
        off[k] = (u32)q_offsets[k];
        for (i64 i = q_offsets[k]; i < q_offsets[k + 1]; i++) end[i] = (u32)q_offsets[k + 1];
    }
    off[n_kp] = n_q;
    // the score walk's workgroups take whole keyphrases (score.h: score_walk_kernel, blk): consecutive keyphrases packed
    // into stretches of at most BLOCK suffixes
    std::vector<u32> blk;
    blk.push_back(0);
    bool fits = kn.score_fused;
    for (int32_t k = 0, used = 0; k < n_kp && fits; k++) {
        const i64 len = q_offsets[k + 1] - q_offsets[k];
        if (len > BLOCK) { fits = false; break; }
        if (used + len > BLOCK) { blk.push_back((u32)k); used = 0; }
        used += (int32_t)len;
    }
    blk.push_back((u32)n_kp);
    h->n_blk = fits ? (u32)blk.size() - 1 : 0;
    if (fits) HIP_CHECK(hipMemcpyAsync(h->q_blk, blk.data(), blk.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipMemcpyAsync(h->q_raw, q_symbols, (size_t)n_q * 4, hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipMemcpyAsync(h->q_end, end.data(), (size_t)n_q * 4, hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipMemcpyAsync(h->q_off, off.data(), off.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    h->n_kp = (u32)n_kp;
    h->n_q = n_q;
    h->score_chunk = chunk;
}

// device memory for n_docs rows of bins + 1 entries plus the fill's chunk scratch; false if it cannot be had
static bool kgram_reserve(east_hip_index *h, u64 bins, u32 n_docs)
{
    // (room for the pair layout: 8-byte entries of the last level + the table of the level above; the filled 4-byte
    // layout with its chunk scratch is smaller)
    const size_t chunks = (size_t)((bins + KGF_CHUNK - 1) / KGF_CHUNK);
    const size_t bytes = (2 * (size_t)(bins + 1) + (size_t)(bins / 2 + 2) + (size_t)(bins / 4 + 8) + 2 * chunks) * n_docs * 4 + 256;
    if (bytes <= h->kg_cap) return true;
    HIP_CHECK(hipStreamSynchronize(h->stream));
    if (h->kg) HIP_CHECK(hipFree(h->kg));
    h->kg = nullptr;
    h->kg_cap = 0;
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
    h->kg = (u32 *)p;
    h->kg_cap = bytes;
    return true;
}

// k-gram bucket tables of the current index (score.h); k = 0 when the alphabet is too wide
static void ensure_kgram(east_hip_index *h, Ctx &ctx)
{
    if (h->kg_built) return;
    if (h->kg_marked && h->kg_pairs) {
        // the pair layout: the last level stays as the build marked it (+ its end entries), the small table above it is filled
        const u32 bins3 = h->kg_bins / h->kg_A;
        LAUNCH(ctx, kgram_pairs_end_kernel, ceil_div_u32(h->n_docs, BLOCK), (const u32 *)h->doc_off, h->n_docs, h->kg_bins, h->kg);
        LAUNCH(ctx, kgram_fill_kernel, h->n_docs, (const u32 *)h->doc_off, bins3, h->kg3);
        // (levels 1 .. k - 2 as tables of their own: A + 1, A^2 + 1, ... entries per document)
        h->kg_up_stride = 0;
        u32 len = h->kg_A;
        for (int l = 1; l < h->kg_k - 1; l++) { h->kg_up_stride += len + 1; len *= h->kg_A; }
        if (h->kg_up_stride && h->kg_up)
            LAUNCH(ctx, kgram_upper_kernel, h->n_docs, (const u32 *)h->kg3, bins3, h->kg_A, h->kg_k - 1, h->kg_up_stride, h->kg_up);
        h->kg_built = true;
        return;
    }
    if (h->kg_marked) {
        // the build left the bucket starts in the table: suffix minimum per document, in chunks
        const u32 bins = h->kg_bins, n_chunks = ceil_div_u32(bins, KGF_CHUNK);
        u32 *cmin = h->kg + (size_t)(bins + 1) * h->n_docs, *csuf = cmin + (size_t)n_chunks * h->n_docs;
        LAUNCH(ctx, kgram_chunk_min_kernel, dim3(n_chunks, h->n_docs), (const u32 *)h->kg, bins, n_chunks, cmin);
        LAUNCH(ctx, kgram_chunk_suffix_kernel, h->n_docs, (const u32 *)cmin, (const u32 *)h->doc_off, n_chunks, csuf);
        LAUNCH(ctx, kgram_chunk_fill_kernel, dim3(n_chunks, h->n_docs), (const u32 *)csuf, (const u32 *)h->doc_off, bins,
               n_chunks, h->kg);
        h->kg_built = true;
        return;
    }
    h->kg_k = 0;
    h->kg_pairs = false;
    h->kg_built = true;
    if (!h->use_s8 || h->n_docs > 65535) return;
    const u32 A = h->sigma_t + 2;
    int k = 0;
    u64 bins = 1;
    while (k < KGRAM_MAX_K && bins * A <= KGRAM_MAX_BINS && bins * A * 16 <= h->n / h->n_docs + 4096 &&
           (bins * A + 1) * h->n_docs * 4 <= ((u64)1 << 30)) {
        bins *= A;
        k++;
    }
    if (k == 0) return;
    const size_t bytes = (size_t)(bins + 1) * h->n_docs * 4;
    if (!kgram_reserve(h, bins, h->n_docs)) return;              // no table: plain binary search
    if ((u64)h->n / h->n_docs >= 256 * bins) {
        // long documents: every table entry by binary search on the suffix array
        LAUNCH(ctx, kgram_search_kernel, dim3(ceil_div_u32(bins + 1, BLOCK), h->n_docs), (const u32 *)h->sa,
               (const uint8_t *)h->s8, (const u32 *)h->doc_off, k, A, (u32)bins, h->kg);
    } else {
        HIP_CHECK(hipMemsetAsync(h->kg, 0xFF, bytes, h->stream));
        i64 longest = 0;
        for (u32 d = 0; d < h->n_docs; d++) longest = std::max(longest, h->h_doc_off[d + 1] - h->h_doc_off[d]);
        if (h->n_docs > 1)
            LAUNCH_NAMED(ctx, "kgram_mark_kernel", kgram_mark_tiled_kernel,
                         dim3(ceil_div_u32((u64)longest + 3, BLOCK * 4), h->n_docs), (const u32 *)h->lcp, (const u32 *)h->sa,
                         (const uint8_t *)h->s8, (const u32 *)h->doc_off, h->n_docs, h->n, k, A, (u32)bins, h->kg);
        else
            LAUNCH(ctx, kgram_mark_kernel, dim3(ceil_div_u32((u64)longest, BLOCK), h->n_docs), (const u32 *)h->lcp,
                   (const u32 *)h->sa, (const uint8_t *)h->s8, (const u32 *)h->doc_off, h->n_docs, h->n, k, A, (u32)bins,
                   h->kg);
        LAUNCH(ctx, kgram_fill_kernel, h->n_docs, (const u32 *)h->doc_off, (u32)bins, h->kg);
    }
    h->kg_k = k;
    h->kg_A = A;
    h->kg_bins = (u32)bins;
}

// queues the score kernels; result in h->table (K x D) / h->suffix (D x S)
static void score_resident(east_hip_index *h, int normalized, unsigned long long *probe_count = nullptr,
                           double *suffix_host = nullptr)
{
    if (!h) east_throw(EAST_HIP_ERR_INVALID, "null handle");
    if (!h->built) east_throw(EAST_HIP_ERR_NOT_BUILT, "no index has been built on this handle");
    if (!h->n_kp) east_throw(EAST_HIP_ERR_INVALID, "no keyphrases set");
    use_device(h);
    Ctx ctx;
    ctx.stream = h->stream;
    ctx.prof = &h->prof;
    HIP_CHECK(hipEventRecord(h->ev0, h->stream));
    ensure_kgram(h, ctx);
    LAUNCH(ctx, query_map_kernel, ceil_div_u32(h->n_q, BLOCK), (const u32 *)h->q_raw, h->n_q,
           (const u32 *)h->code_map, (const u32 *)h->hi_bits, (const u32 *)h->hi_rank,
           h->sigma_hi ? h->sigma_t - h->sigma_hi + 1u : 0u, h->q_code);
    KgTables kt;
    kt.kg = h->kg; kt.kg3 = h->kg3; kt.k = h->kg_k; kt.pairs = h->kg_k > 0 && h->kg_pairs; kt.A = h->kg_A; kt.bins = h->kg_bins;
    if (kt.pairs && h->kg_up_stride && h->kg_up) {
        kt.up = h->kg_up;
        kt.up_stride = h->kg_up_stride;
        static const bool up_lds_off = getenv("EAST_HIP_SCORE_UP_LDS") && atoi(getenv("EAST_HIP_SCORE_UP_LDS")) == 0;   // (A/B timing)
        kt.up_lds = !up_lds_off && h->kg_up_stride <= KG_UP_LDS_WORDS;
    }
    if (kt.k > 0) kt.finish();
    kt.endgame = ctx.knobs.score_endgame;
    // whole keyphrases per workgroup, summed in the walk (no per-suffix results unless the caller wants them: then the
    // documents go a stretch at a time, as far as the scratch reaches); otherwise per-suffix results + the reduction kernel
    const bool fused = h->n_blk > 0;
    u32 chunk = h->score_chunk;
    if (fused && !suffix_host) {
        // (no scratch to bound the stretch -- the grid does: a launch of at most Knobs::score_grid_blocks workgroups, far below
        // HIP's limit of 2^32 threads per grid dimension; many short documents times thousands of keyphrases go a stretch
        // of documents at a time, a multiple of 8 for the XCD-aware order)
        const u64 fit = ctx.knobs.score_grid_blocks / h->n_blk;
        chunk = (u32)std::min<u64>(h->n_docs, fit >= 8 ? fit & ~(u64)7 : std::max<u64>(fit, 1));
    }
    for (u32 first = 0; first < h->n_docs; first += chunk) {
        const u32 count = std::min(chunk, h->n_docs - first);
        const int xcd_order = count >= 64;                // see score_walk_kernel
        const u32 per_doc = fused ? h->n_blk : ceil_div_u32(h->n_q, BLOCK);
        const u64 walk_grid64 = (u64)(xcd_order ? 8u * ceil_div_u32(count, 8) : count) * per_doc;
        if (walk_grid64 >= ((u64)1 << 24)) east_throw(EAST_HIP_ERR_INVALID, "keyphrase set too large for one score launch");
        const u32 walk_grid = (u32)walk_grid64;
        double *suffix = fused && !suffix_host ? (double *)nullptr : h->suffix;
        const u32 *blk = fused ? (const u32 *)h->q_blk : (const u32 *)nullptr;
        if (h->use_s8)
            LAUNCH_NAMED(ctx, "score_walk_kernel", (score_walk_kernel<uint8_t>), walk_grid, (const uint8_t *)h->s8,
                         (const u32 *)h->sa, (const u32 *)h->doc_off, (const u32 *)h->n_strings, h->n_docs,
                         (const u32 *)h->q_code, (const u32 *)h->q_end, h->n_q, normalized, kt, xcd_order, first, count, suffix,
                         probe_count, blk, h->n_blk, (const u32 *)h->q_off, h->table);
        else
            LAUNCH_NAMED(ctx, "score_walk_kernel", (score_walk_kernel<u32>), walk_grid, (const u32 *)h->s,
                         (const u32 *)h->sa, (const u32 *)h->doc_off, (const u32 *)h->n_strings, h->n_docs,
                         (const u32 *)h->q_code, (const u32 *)h->q_end, h->n_q, normalized, kt, xcd_order, first, count, suffix,
                         probe_count, blk, h->n_blk, (const u32 *)h->q_off, h->table);
        if (!fused)
            LAUNCH(ctx, score_reduce_kernel, ceil_div_u32((u64)h->n_kp * count, BLOCK), (const double *)h->suffix,
                   (const u32 *)h->q_off, h->n_kp, h->n_docs, h->n_q, first, count, h->table);
        if (suffix_host)                                  // the per-suffix results of this stretch of documents (D x S, row-major)
            HIP_CHECK(hipMemcpyAsync(suffix_host + (size_t)first * h->n_q, h->suffix, (size_t)count * h->n_q * 8,
                                     hipMemcpyDeviceToHost, h->stream));
    }
    HIP_CHECK(hipEventRecord(h->ev1, h->stream));
}

// ------------------------------------------------------------------ C ABI --
template <class F> static int guarded(F f)
{
    int rc = EAST_HIP_OK;
    try {
        f();
    } catch (const EastError &e) {
        g_last_error = e.msg;
        rc = e.code;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        rc = EAST_HIP_ERR_INTERNAL;
    }
    restore_device();
    return rc;
}

extern "C" {

const char *east_hip_version(void) { return "east-hip 0.1 (gfx950)"; }
const char *east_hip_last_error(void) { return g_last_error.c_str(); }

int east_hip_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) {
        g_last_error = "no HIP device available (this library has no CPU fallback)";
        return EAST_HIP_ERR_NO_DEVICE;
    }
    return c;
}

int east_hip_create(int device, int64_t reserve_symbols, east_hip_handle_t *out)
{
    if (out) *out = nullptr;
    return guarded([&] {
        if (!out) east_throw(EAST_HIP_ERR_INVALID, "null out pointer");
        int c = 0;
        if (hipGetDeviceCount(&c) != hipSuccess || c <= 0)
            east_throw(EAST_HIP_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
        if (device < 0 || device >= c) east_throw(EAST_HIP_ERR_NO_DEVICE, "device ordinal out of range");
        if (reserve_symbols < 0 || reserve_symbols >= (i64)0x7FFFFFF0)
            east_throw(EAST_HIP_ERR_INVALID, "reserve_symbols out of range");
        east_hip_index *h = new east_hip_index();
        h->device = device;
        try {
            use_device_ordinal(device);
            HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
            HIP_CHECK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));      // (the streamed text preparation's uploads)
            HIP_CHECK(hipEventCreate(&h->ev0));
            HIP_CHECK(hipEventCreate(&h->ev1));
            void *g = nullptr;
            HIP_CHECK(hipMalloc(&g, (TEXT_SYMBOLS + PRESENT_WORDS) * sizeof(u32)));
            h->guess = (u32 *)g;
            if (reserve_symbols > 0) {
                // (a handle made for builds large enough to go up narrowed -- upload_symbols_narrow -- reserves the narrow
                // staging too: its first east_hip_build does not find the arena a few MB short and grow it.  The upload ring is
                // NOT pinned here: with it in place a fresh handle's first device-resident build measured 3 % slower -- 1.93
                // against 1.86 ms wall for the 64 MiB document, with the ring pinned in the background 3.2 --; the first
                // host-resident build of that size pins it, build_common)
                const bool narrow_size = (u64)reserve_symbols >= SYM_NARROW_MIN && getenv("EAST_HIP_NO_SYMBOL_NARROW") == nullptr;
                const size_t narrow_bytes = narrow_size ? (((size_t)reserve_symbols + 8) * 2 + 255) & ~(size_t)255 : 0;
                ensure_arena(h, plan_arena_bytes((u32)reserve_symbols, 1) + (((size_t)reserve_symbols * 4 + 255) & ~(size_t)255) + narrow_bytes);
            }
        } catch (...) {
            east_hip_destroy(h);
            throw;
        }
        *out = h;
    });
}

void east_hip_destroy(east_hip_handle_t h)
{
    if (!h) return;
    int cur = -1;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->arena.base) (void)hipFree(h->arena.base);
    if (h->q_buf) (void)hipFree(h->q_buf);
    if (h->kg) (void)hipFree(h->kg);
    if (h->prep_sym) (void)hipFree(h->prep_sym);
    if (h->tp_tables) (void)hipFree(h->tp_tables);
    if (h->ht_tab) (void)hipFree(h->ht_tab);
    for (auto e : h->copy_events) (void)hipEventDestroy(e);
    ring_adopt(h, true);
    for (auto e : h->ring_events) (void)hipEventDestroy(e);
    if (h->ring) (void)hipHostFree(h->ring);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->guess) (void)hipFree(h->guess);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (cur >= 0 && cur != h->device) (void)hipSetDevice(cur);
    delete h;
}

int east_hip_build(east_hip_handle_t h, const uint32_t *symbols, int64_t n_total, const int64_t *doc_offsets,
                   const int32_t *n_strings, int32_t n_docs)
{
    return guarded([&] { build_common(h, symbols, true, n_total, doc_offsets, n_strings, n_docs, h && h->tagged_input); });
}

int east_hip_build_device(east_hip_handle_t h, const uint32_t *d_symbols, int64_t n_total,
                          const int64_t *doc_offsets, const int32_t *n_strings, int32_t n_docs)
{
    return guarded([&] { build_common(h, d_symbols, false, n_total, doc_offsets, n_strings, n_docs, h && h->tagged_input); });
}

int east_hip_set_symbol_encoding(east_hip_handle_t h, int32_t encoding)
{
    return guarded([&] {
        if (!h) east_throw(EAST_HIP_ERR_INVALID, "null handle");
        if (encoding != EAST_HIP_SYMBOLS_REFERENCE && encoding != EAST_HIP_SYMBOLS_TAGGED)
            east_throw(EAST_HIP_ERR_INVALID, "unknown symbol encoding");
        h->tagged_input = encoding == EAST_HIP_SYMBOLS_TAGGED;
    });
}

int east_hip_prepared_encoding(east_hip_handle_t h)
{
    return h && h->prep_tagged ? EAST_HIP_SYMBOLS_TAGGED : EAST_HIP_SYMBOLS_REFERENCE;
}

int east_hip_build_texts(east_hip_handle_t h, const uint8_t *bytes, int64_t n_bytes, const int64_t *text_offsets,
                         int32_t n_docs, const uint8_t *cp_class, const uint32_t *cp_upper, const uint32_t *word_hi,
                         const uint32_t *digit_hi, const uint32_t *hi_upper_from, const uint32_t *hi_upper_to,
                         int32_t n_hi_upper)
{
    return guarded([&] {
        build_from_texts(h, bytes, n_bytes, text_offsets, n_docs, cp_class, cp_upper, word_hi, digit_hi, hi_upper_from,
                         hi_upper_to, n_hi_upper);
    });
}

int east_hip_build_texts_v(east_hip_handle_t h, const uint8_t *const *texts, const int64_t *lengths, int32_t n_docs,
                           const uint8_t *cp_class, const uint32_t *cp_upper, const uint32_t *word_hi,
                           const uint32_t *digit_hi, const uint32_t *hi_upper_from, const uint32_t *hi_upper_to,
                           int32_t n_hi_upper)
{
    return guarded([&] {
        if (!texts || !lengths || n_docs < 1) east_throw(EAST_HIP_ERR_INVALID, "null argument or no documents");
        std::vector<i64> off((size_t)n_docs + 1, 0);
        for (int32_t d = 0; d < n_docs; d++) {
            if (lengths[d] < 0) east_throw(EAST_HIP_ERR_INVALID, "negative text length");
            off[d + 1] = off[d] + lengths[d] + 1;                                   // + the separator
        }
        build_from_texts(h, nullptr, off[n_docs], off.data(), n_docs, cp_class, cp_upper, word_hi, digit_hi, hi_upper_from,
                         hi_upper_to, n_hi_upper, texts);
    });
}

int east_hip_get_prepared(east_hip_handle_t h, int64_t *n_total, int64_t *doc_offsets, int32_t *n_strings,
                          uint32_t *symbols)
{
    return guarded([&] {
        if (!h) east_throw(EAST_HIP_ERR_INVALID, "null handle");
        if (!h->prep_sym || h->prep_doc_off.empty()) east_throw(EAST_HIP_ERR_NOT_BUILT, "no texts have been prepared on this handle");
        use_device(h);
        if (n_total) *n_total = h->prep_n;
        if (doc_offsets) memcpy(doc_offsets, h->prep_doc_off.data(), h->prep_doc_off.size() * sizeof(i64));
        if (n_strings) memcpy(n_strings, h->prep_n_strings.data(), h->prep_n_strings.size() * sizeof(int32_t));
        if (symbols) {
            HIP_CHECK(hipMemcpyAsync(symbols, h->prep_sym, (size_t)h->prep_n * 4, hipMemcpyDeviceToHost, h->stream));
            HIP_CHECK(hipStreamSynchronize(h->stream));
        }
    });
}

double east_hip_last_prep_ms(east_hip_handle_t h) { return h ? (double)h->last_prep_ms : -1.0; }

int east_hip_get_tables(east_hip_handle_t h, int32_t doc, int32_t *suftab, int32_t *lcptab, int32_t *anntab,
                        int32_t *childtab_up, int32_t *childtab_down, int32_t *childtab_next_l_index)
{
    return guarded([&] {
        if (!h) east_throw(EAST_HIP_ERR_INVALID, "null handle");
        if (!h->built) east_throw(EAST_HIP_ERR_NOT_BUILT, "no index has been built on this handle");
        if (doc < 0 || (u32)doc >= h->n_docs) east_throw(EAST_HIP_ERR_INVALID, "document index out of range");
        use_device(h);
        if ((childtab_up || childtab_down || childtab_next_l_index) && !h->child_built) {
            Ctx ctx;
            ctx.stream = h->stream;
            ctx.prof = &h->prof;
            // (the lists of the ranks the streaming pass leaves to the pyramid live in the arena's temporary region)
            ctx.arena = &h->arena;
            const size_t mark = h->arena.mark();
            const u32 n_tiles = ceil_div_u32(h->n, CH_TILE);
            u32 *wide_list = h->arena.alloc<u32>((size_t)n_tiles * CH_TILE);
            u32 *wide_count = h->arena.alloc<u32>(n_tiles);
            LAUNCH(ctx, child_stream_kernel, n_tiles, h->pyr, (const u32 *)h->doc_off, h->n_docs, h->n, h->up, h->down,
                   h->next, wide_list, wide_count);
            LAUNCH(ctx, child_wide_kernel, ceil_div_u32(n_tiles, BLOCK / CH_WIDE_SLOTS), h->pyr, (const u32 *)h->doc_off,
                   h->n_docs, h->n, n_tiles, (const u32 *)wide_list, (const u32 *)wide_count, h->up, h->down, h->next);
            h->arena.release(mark);
            h->child_built = true;
        }
        const size_t b = (size_t)h->h_doc_off[doc], nd = (size_t)(h->h_doc_off[doc + 1] - h->h_doc_off[doc]);
        struct { int32_t *dst; const u32 *src; } jobs[6] = {
            {suftab, h->sa}, {lcptab, h->lcp}, {anntab, h->ann},
            {childtab_up, h->up}, {childtab_down, h->down}, {childtab_next_l_index, h->next}};
        for (auto &j : jobs)
            if (j.dst) HIP_CHECK(hipMemcpyAsync(j.dst, j.src + b, nd * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        if (suftab)
            for (size_t i = 0; i < nd; i++) suftab[i] -= (int32_t)b;      // global -> document-local positions
    });
}

int east_hip_set_keyphrases(east_hip_handle_t h, const uint32_t *q_symbols, const int64_t *q_offsets,
                            int32_t n_keyphrases)
{
    return guarded([&] { set_keyphrases(h, q_symbols, q_offsets, n_keyphrases); });
}

int east_hip_score_resident(east_hip_handle_t h, int normalized, double *d_out)
{
    return guarded([&] {
        score_resident(h, normalized);
        if (d_out)
            HIP_CHECK(hipMemcpyAsync(d_out, h->table, (size_t)h->n_kp * h->n_docs * 8, hipMemcpyDeviceToDevice,
                                     h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        HIP_CHECK(hipEventElapsedTime(&h->last_score_ms, h->ev0, h->ev1));
    });
}

int east_hip_score_probes(east_hip_handle_t h, int normalized, int64_t *probes)
{
    return guarded([&] {
        if (!h || !probes) east_throw(EAST_HIP_ERR_INVALID, "null handle or output");
        if (!h->n_kp) east_throw(EAST_HIP_ERR_INVALID, "no keyphrases set");
        use_device(h);
        unsigned long long *d_count = (unsigned long long *)h->q_buf;
        HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), h->stream));
        score_resident(h, normalized, d_count);
        unsigned long long c = 0;
        HIP_CHECK(hipMemcpyAsync(&c, d_count, sizeof(c), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        *probes = (int64_t)c;
    });
}

int east_hip_score_resident_async(east_hip_handle_t h, int normalized)
{
    return guarded([&] { score_resident(h, normalized); });
}

int east_hip_score_table(east_hip_handle_t h, const uint32_t *q_symbols, const int64_t *q_offsets,
                         int32_t n_keyphrases, int normalized, double *out, double *suffix_out)
{
    return guarded([&] {
        if (!out) east_throw(EAST_HIP_ERR_INVALID, "null output table");
        set_keyphrases(h, q_symbols, q_offsets, n_keyphrases);
        score_resident(h, normalized, nullptr, suffix_out);
        HIP_CHECK(hipMemcpyAsync(out, h->table, (size_t)h->n_kp * h->n_docs * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        HIP_CHECK(hipEventElapsedTime(&h->last_score_ms, h->ev0, h->ev1));
    });
}

int east_hip_score_table_grouped(east_hip_handle_t h, const uint32_t *q_symbols, const int64_t *q_offsets,
                                 int32_t n_queries, const int64_t *group_offsets, int32_t n_groups, int normalized,
                                 double *out)
{
    return guarded([&] {
        if (!out || !group_offsets || n_groups < 1) east_throw(EAST_HIP_ERR_INVALID, "null output table or no groups");
        if (group_offsets[0] != 0 || group_offsets[n_groups] != n_queries)
            east_throw(EAST_HIP_ERR_INVALID, "group_offsets must start at 0 and end at n_queries");
        for (int32_t g = 0; g < n_groups; g++)
            if (group_offsets[g + 1] <= group_offsets[g]) east_throw(EAST_HIP_ERR_INVALID, "empty group");
        set_keyphrases(h, q_symbols, q_offsets, n_queries);
        std::vector<u32> goff((size_t)n_groups + 1);
        for (int32_t g = 0; g <= n_groups; g++) goff[g] = (u32)group_offsets[g];
        HIP_CHECK(hipMemcpyAsync(h->group_off, goff.data(), goff.size() * 4, hipMemcpyHostToDevice, h->stream));
        score_resident(h, normalized);
        Ctx ctx;
        ctx.stream = h->stream;
        ctx.prof = &h->prof;
        LAUNCH(ctx, score_group_max_kernel, ceil_div_u32((u64)n_groups * h->n_docs, BLOCK), (const double *)h->table,
               (const u32 *)h->group_off, (u32)n_groups, h->n_docs, h->table_g);
        HIP_CHECK(hipEventRecord(h->ev1, h->stream));
        HIP_CHECK(hipMemcpyAsync(out, h->table_g, (size_t)n_groups * h->n_docs * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));       // (also covers goff)
        HIP_CHECK(hipEventElapsedTime(&h->last_score_ms, h->ev0, h->ev1));
    });
}

int east_hip_get_lcp_intervals(east_hip_handle_t h, int32_t doc, int32_t *left)
{
    return guarded([&] {
        if (!h || !left) east_throw(EAST_HIP_ERR_INVALID, "null handle or output");
        if (!h->built) east_throw(EAST_HIP_ERR_NOT_BUILT, "no index has been built on this handle");
        if (doc < 0 || (u32)doc >= h->n_docs) east_throw(EAST_HIP_ERR_INVALID, "document index out of range");
        use_device(h);
        Ctx ctx;
        ctx.stream = h->stream;
        ctx.prof = &h->prof;
        const u32 seg = (u32)h->h_doc_off[doc], nd = (u32)(h->h_doc_off[doc + 1] - h->h_doc_off[doc]);
        // (scratch from the arena's temporary region, idle between builds: the child tables stay as they are)
        const size_t mark = h->arena.mark();
        u32 *scratch = h->arena.alloc<u32>(nd);
        LAUNCH(ctx, interval_left_kernel, ceil_div_u32(nd, BLOCK), h->pyr, (const u32 *)h->ann, seg, nd, scratch);
        HIP_CHECK(hipMemcpyAsync(left, scratch, (size_t)nd * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->arena.release(mark);
    });
}

int east_hip_reset(east_hip_handle_t h)
{
    return guarded([&] {
        if (!h) east_throw(EAST_HIP_ERR_INVALID, "null handle");
        use_device(h);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->built = false;
        h->n = 0;
        h->n_docs = 0;
        h->n_kp = 0;
        h->n_q = 0;
        h->kg_built = false;
        h->child_built = false;
        h->prep_n = 0;
        h->prep_doc_off.clear();
        h->prep_n_strings.clear();
        h->prep_tagged = false;
        h->tagged_input = false;
        h->sigma_hi = 0;
        h->prof.enabled = false;
        h->prof.only.clear();
        h->plan_wide = h->plan_fused = h->plan_ht = h->plan_persist = -1;
        h->ht_valid = false;
        h->stats = Stats();
        // a recycled handle keeps its stream and a small arena, not gigabytes of side allocations
        const size_t keep = (size_t)64 << 20;
        if (h->q_cap > keep) { (void)hipFree(h->q_buf); h->q_buf = nullptr; h->q_cap = 0; }
        if (h->kg_cap > keep) { (void)hipFree(h->kg); h->kg = nullptr; h->kg_cap = 0; }
        if (h->prep_cap > keep) { (void)hipFree(h->prep_sym); h->prep_sym = nullptr; h->prep_cap = 0; }
    });
}

int east_hip_synchronize(east_hip_handle_t h)
{
    return guarded([&] {
        if (!h) east_throw(EAST_HIP_ERR_INVALID, "null handle");
        use_device(h);
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}

void *east_hip_stream(east_hip_handle_t h) { return h ? (void *)h->stream : nullptr; }

int east_hip_build_info(east_hip_handle_t h, int64_t *out, int32_t cap)
{
    if (!h || !out) return EAST_HIP_ERR_INVALID;
    const int64_t v[27] = {h->n, h->n_docs, h->m_total, h->sigma_t, h->bits0, h->stats.levels,
                           (int64_t)h->arena.cap, (int64_t)h->arena.high, h->stats.radix_passes,
                           h->stats.radix_elems, h->stats.radix_elem_bytes, h->stats.radix_passes_u32,
                           h->stats.radix_elems_u32, h->stats.radix_passes_u64, h->stats.radix_elems_u64,
                           h->stats.levels_resolved, h->stats.merge_elems, h->stats.refine_rounds,
                           h->stats.window_sorted, h->stats.lds_sorted, h->stats.fused_finish, h->stats.first_kept,
                           h->stats.first_n, h->stats.ht_keys, h->stats.seg_sort, h->narrow_upload,
                           h->stats.persist_rounds};
    for (int i = 0; i < 27 && i < cap; i++) out[i] = v[i];
    return 27;
}

int east_hip_profile_enable(east_hip_handle_t h, int on)
{
    if (!h) return EAST_HIP_ERR_INVALID;
    return guarded([&] {
        use_device(h);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->prof.reset();
        h->prof.enabled = on != 0;
    });
}

int east_hip_profile_only(east_hip_handle_t h, const char *kernel)
{
    if (!h) return EAST_HIP_ERR_INVALID;
    h->prof.only = kernel ? kernel : "";
    return EAST_HIP_OK;
}

int64_t east_hip_profile_report(east_hip_handle_t h, char *buf, int64_t cap)
{
    if (!h || !buf || cap < 1) return EAST_HIP_ERR_INVALID;
    int cur = -1;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    h->prof.collect();
    if (cur >= 0 && cur != h->device) (void)hipSetDevice(cur);
    std::string out;
    for (auto &s : h->prof.sums) {
        char line[256];
        snprintf(line, sizeof(line), "%s\t%lld\t%.6f\n", s.name.c_str(), (long long)s.count, s.ms);
        out += line;
    }
    const int64_t nb = (int64_t)out.size() < cap - 1 ? (int64_t)out.size() : cap - 1;
    memcpy(buf, out.data(), (size_t)nb);
    buf[nb] = 0;
    return (int64_t)out.size();
}

int64_t east_hip_plan_arena_bytes(int64_t n_total, int32_t n_docs)
{
    if (n_total < 1 || n_total >= (i64)0x7FFFFFF0 || n_docs < 1) return EAST_HIP_ERR_INVALID;
    int64_t r = EAST_HIP_ERR_INTERNAL;
    guarded([&] { r = (int64_t)plan_arena_bytes((u32)n_total, (u32)n_docs); });
    return r;
}

int64_t east_hip_plan_arena_bytes_lean(int64_t n_total, int32_t n_docs)
{
    if (n_total < 1 || n_total >= (i64)0x7FFFFFF0 || n_docs < 1) return EAST_HIP_ERR_INVALID;
    int64_t r = EAST_HIP_ERR_INTERNAL;
    guarded([&] { r = (int64_t)plan_arena_bytes((u32)n_total, (u32)n_docs, true); });
    return r;
}

double east_hip_last_build_ms(east_hip_handle_t h) { return h ? (double)h->last_build_ms : -1.0; }
double east_hip_last_score_ms(east_hip_handle_t h) { return h ? (double)h->last_score_ms : -1.0; }

}  // extern "C"

// ---- kernel-level test entry points --------------------------------------------
struct DebugScope {
    hipStream_t stream = nullptr;
    Arena arena;
    Stats stats;
    Ctx ctx;
    DebugScope(int device, size_t bytes)
    {
        int c = 0;
        if (hipGetDeviceCount(&c) != hipSuccess || c <= 0)
            east_throw(EAST_HIP_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
        if (device < 0 || device >= c) east_throw(EAST_HIP_ERR_NO_DEVICE, "device ordinal out of range");
        use_device_ordinal(device);
        HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        void *p = nullptr;
        HIP_CHECK(hipMalloc(&p, bytes));
        arena.base = (char *)p;
        arena.cap = bytes;
        ctx.stream = stream;
        ctx.arena = &arena;
        ctx.stats = &stats;
    }
    ~DebugScope()
    {
        if (stream) (void)hipStreamSynchronize(stream);
        if (arena.base) (void)hipFree(arena.base);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

template <class K> static void debug_sort(int device, K *keys, u32 *vals, i64 n, int bits)
{
    if (n < 0 || n >= (i64)0x7FFFFFF0 || !keys || !vals || bits < 1 || bits > (int)sizeof(K) * 8)
        east_throw(EAST_HIP_ERR_INVALID, "bad radix sort arguments");
    if (n == 0) return;
    DebugScope sc(device, (size_t)n * (sizeof(K) + 4) * 2 + (40u << 20));
    SortBufs<K> sb;
    for (int k = 0; k < 2; k++) { sb.keys[k] = sc.arena.alloc<K>(n); sb.vals[k] = sc.arena.alloc<u32>(n); }
    HIP_CHECK(hipMemcpyAsync(sb.keys[0], keys, (size_t)n * sizeof(K), hipMemcpyHostToDevice, sc.stream));
    HIP_CHECK(hipMemcpyAsync(sb.vals[0], vals, (size_t)n * 4, hipMemcpyHostToDevice, sc.stream));
    const int r = radix_sort_pairs<K>(sc.ctx, sb, (u32)n, bits);
    HIP_CHECK(hipMemcpyAsync(keys, sb.keys[r], (size_t)n * sizeof(K), hipMemcpyDeviceToHost, sc.stream));
    HIP_CHECK(hipMemcpyAsync(vals, sb.vals[r], (size_t)n * 4, hipMemcpyDeviceToHost, sc.stream));
    HIP_CHECK(hipStreamSynchronize(sc.stream));
}

extern "C" {

int east_hip_debug_set_rank_bucket_bytes(int64_t bytes)
{
    if (bytes < 0) return EAST_HIP_ERR_INVALID;
    knobs_update([&](Knobs &k) { k.rank_bucket_bytes = (size_t)bytes; k.plan_epoch++; });
    return EAST_HIP_OK;
}

int east_hip_debug_set_window_sort(int enabled)
{
    // 0: DC3 only; 1: the default; 2: lean (no refinement rounds); 3: 64-bit window keys; 4 / 5: as 1 / 3 without the
    // fused finish (every radix pass global, then lvl0_place_kernel)
    knobs_update([&](Knobs &k) {
    k.window_sort = enabled != 0;
    k.force_lean = enabled == 2;
    k.force_wide_keys = enabled == 3 || enabled == 5;
    k.fused_finish = enabled != 4 && enabled != 5 && enabled != 9 && getenv("EAST_HIP_NO_FUSED_FINISH") == nullptr;
    k.force_fused = enabled == 6 || getenv("EAST_HIP_FORCE_FUSED") != nullptr;                        // 6: as 1, the fused finish whatever the plan says (skewed text through it)
    // 7: as 1, first-level keys of variable-length code words wherever a code can be made (ht_code.h); 9: the same
    // without the fused finish; 8: as 1 without such keys
    k.ht_mode = enabled == 7 || enabled == 9 ? 1 : enabled == 8 ? 0 : env_int("EAST_HIP_HT", -1);
    k.plan_epoch++;
    });
    return EAST_HIP_OK;
}

int east_hip_debug_set_segmented_sort(int mode)
{
    // -1: the default (by size: a few large documents); 0: never (the document number is a key digit); 1: wherever it
    // can be done (2 .. RS_SEG_MAX_DOCS documents of any size)
    knobs_update([&](Knobs &k) { k.seg_mode = mode < 0 ? env_int("EAST_HIP_SEG", -1) : (mode != 0); k.plan_epoch++; });
    return EAST_HIP_OK;
}

int east_hip_debug_alphabetic_code(const uint64_t *weights, int32_t n, uint32_t *code, int32_t *len)
{
    // host only: the order-preserving variable-length code of csrc/ht_code.h for n symbols with the given weights
    if (!weights || !code || !len || n < 1) return EAST_HIP_ERR_INVALID;
    std::vector<u64> w(weights, weights + n);
    std::vector<u32> c;
    std::vector<int> l;
    if (!ht_build_code(w, c, l)) return EAST_HIP_ERR_DOMAIN;
    for (int i = 0; i < n; i++) { code[i] = c[i]; len[i] = l[i]; }
    return EAST_HIP_OK;
}

int east_hip_debug_narrow_symbols(const uint32_t *symbols, int64_t n, uint16_t *out, int vector)
{
    // host only: what upload_symbols_narrow's host threads do to a stretch of symbols on its way into the pinned ring
    // (vector != 0: the AVX2 form where the CPU has it; 0: the plain loop)
    if (!symbols || !out || n < 0) return EAST_HIP_ERR_INVALID;
    if (vector) narrow_symbols(symbols, out, (size_t)n);
    else
        for (int64_t i = 0; i < n; i++) out[i] = symbols[i] < TEXT_SYMBOLS ? (uint16_t)symbols[i] : (uint16_t)SYM_TERMINATOR16;
    return vector && g_have_avx2 ? 1 : EAST_HIP_OK;
}

int east_hip_debug_narrow_symbols8(const uint32_t *symbols, int64_t n, uint8_t *out, int vector)
{
    // host only, as above: the narrowing to bytes.  Returns 1 when every symbol fitted (text below 0xFF, terminators from
    // U+0A00 on), 0 when one did not, + 2 when the AVX2 form ran
    if (!symbols || !out || n < 0) return EAST_HIP_ERR_INVALID;
    bool ok = true;
    if (vector) ok = narrow_symbols8(symbols, out, (size_t)n);
    else
        for (int64_t i = 0; i < n; i++) { const u32 c = symbols[i]; ok &= !(c >= 0xFFu && c < TEXT_SYMBOLS); out[i] = c < 0xFFu ? (uint8_t)c : (uint8_t)0xFFu; }
    return (ok ? 1 : 0) + (vector && g_have_avx2 ? 2 : 0);
}

int east_hip_debug_set_lds_rounds(int enabled)
{
    // 0: every round through the global sort; 1: the default (in-LDS rounds that also classify the next domain, small
    // domains finished by one persistent launch); 2: in-LDS rounds with the stand-alone classification pass, launch by
    // launch; 3: as 1, launch by launch (no persistent kernel)
    knobs_update([&](Knobs &k) {
        k.lds_rounds = enabled != 0;
        k.fused_classify = enabled != 2 && getenv("EAST_HIP_NO_FUSED_CLASSIFY") == nullptr;
        k.persist = enabled == 1 && getenv("EAST_HIP_NO_PERSIST") == nullptr;
    });
    return EAST_HIP_OK;
}

int east_hip_debug_set_persist(int force_large, int max_workgroups)
{
    // the persistent rounds (persist_rounds.h): force_large != 0 -- the large form (tiles' state in global memory, several
    // tiles per workgroup) also where the resident form would do; max_workgroups > 0 -- a grid of at most that many
    // workgroups (0: what the device holds).  (0, 0) = the default.
    knobs_update([&](Knobs &k) {
        k.persist_force_large = force_large != 0;
        k.persist_max_wgs = max_workgroups > 0 ? max_workgroups : 0;
    });
    return EAST_HIP_OK;
}

int east_hip_debug_set_score_scratch(int64_t bytes)
{
    knobs_update([&](Knobs &k) { k.score_scratch_bytes = bytes > 0 ? (size_t)bytes : SCORE_SCRATCH_BYTES; });
    return EAST_HIP_OK;
}

int east_hip_debug_set_score_grid(int64_t workgroups)
{
    // workgroups a launch of the score walk may have when the sums run inside it (0 or less: the default, 2^22)
    knobs_update([&](Knobs &k) { k.score_grid_blocks = workgroups > 0 ? (u64)std::min<int64_t>(workgroups, (int64_t)1 << 23) : SCORE_GRID_BLOCKS; });
    return EAST_HIP_OK;
}

int east_hip_debug_set_text_ring(int mode, int64_t slot_bytes)
{
    // mode -1: separate texts go up through the pinned ring when there are four or more of less than 8 MiB on average
    // (and the preparation is streamed); 0: never; 1: whenever the texts lie apart.  slot_bytes: size of a ring slot
    // (0 or less: the default, 8 MiB; at most that)
    knobs_update([&](Knobs &k) {
        k.tp_ring = mode;
        k.tp_ring_slot = slot_bytes > 0 ? (size_t)std::min<int64_t>(slot_bytes, (int64_t)TP_RING_SLOT) : TP_RING_SLOT;
    });
    return EAST_HIP_OK;
}

int east_hip_debug_set_text_stream(int64_t chunk_bytes)
{
    // -1: the default (inputs of 8 MiB or more go up and are prepared in about five chunks); 0: the raw text goes up and
    // is prepared in one piece; > 0: always in chunks of about that many bytes
    knobs_update([&](Knobs &k) { k.tp_stream = chunk_bytes; });
    return EAST_HIP_OK;
}

int east_hip_debug_set_score_path(int mode)
{
    // 1: the default; 0: the walk as rounds 1-3 ran it -- one filled k-gram table of 4-byte entries, per-suffix results in
    // HBM and a reduction kernel; 2: pair tables, separate reduction; 3: filled table, the sums inside the walk.
    // (takes effect with the next build / the next set of keyphrases)
    knobs_update([&](Knobs &k) {
        k.kg_pairs = mode == 1 || mode == 2 || mode == 4 || mode == 5;
        k.kg_pairs_forced = mode == 4;                  // 4: as 1, the pair tables also for collections of fewer than 16 documents
        k.score_fused = mode == 1 || mode == 3 || mode == 4 || mode == 5;
        k.score_endgame = mode == 5 ? 0 : 1;                 // 5: as 1, binary search down to the last suffix (no register endgame)
    });
    return EAST_HIP_OK;
}

int east_hip_debug_set_speculation(int enabled)
{
    knobs_update([&](Knobs &k) { k.speculate = enabled != 0; });
    return EAST_HIP_OK;
}

int east_hip_debug_radix_sort_u64(int device, uint64_t *keys, uint32_t *vals, int64_t n, int bits)
{
    return guarded([&] { debug_sort<u64>(device, keys, vals, n, bits); });
}

int east_hip_debug_radix_sort_u32(int device, uint32_t *keys, uint32_t *vals, int64_t n, int bits)
{
    return guarded([&] { debug_sort<u32>(device, keys, vals, n, bits); });
}

int east_hip_debug_exclusive_scan(int device, const uint32_t *in, uint32_t *out, int64_t n)
{
    return guarded([&] {
        if (n < 0 || n >= (i64)0x7FFFFFF0 || !in || !out) east_throw(EAST_HIP_ERR_INVALID, "bad scan arguments");
        if (n == 0) return;
        DebugScope sc(device, (size_t)n * 8 + (size_t)ceil_div_u32(n, SCAN_TILE) * 16 + (8u << 20));
        u32 *d_in = sc.arena.alloc<u32>(n), *d_out = sc.arena.alloc<u32>(n);
        HIP_CHECK(hipMemcpyAsync(d_in, in, (size_t)n * 4, hipMemcpyHostToDevice, sc.stream));
        device_scan<ArrIn, false>(sc.ctx, ArrIn{d_in}, (u32)n, d_out);
        HIP_CHECK(hipMemcpyAsync(out, d_out, (size_t)n * 4, hipMemcpyDeviceToHost, sc.stream));
        HIP_CHECK(hipStreamSynchronize(sc.stream));
    });
}

int east_hip_debug_suffix_array(int device, const uint32_t *symbols, int64_t n, uint32_t sigma, int32_t *sa_out,
                                int32_t *levels_out)
{
    return guarded([&] {
        if (n < 1 || n >= (i64)0x7FFFFFF0 || !symbols || !sa_out || sigma < 1)
            east_throw(EAST_HIP_ERR_INVALID, "bad suffix array arguments");
        for (i64 i = 0; i < n; i++)
            if (symbols[i] < 1 || symbols[i] > sigma) east_throw(EAST_HIP_ERR_INVALID, "symbol outside [1, sigma]");
        // measure the arena with the same code path, then run it
        Arena dry;
        dry.dry = true;
        Stats st;
        Ctx dctx;
        dctx.arena = &dry;
        dctx.dry = true;
        dctx.stats = &st;
        (void)dry.alloc<u32>((size_t)n + 3);
        (void)dry.alloc<u32>(n);
        dc3_suffix_array(dctx, nullptr, (u32)n, (u32)std::max<i64>(n, sigma), nullptr);
        DebugScope sc(device, dry.high + (8u << 20));
        u32 *s = sc.arena.alloc<u32>((size_t)n + 3), *sa = sc.arena.alloc<u32>(n);
        HIP_CHECK(hipMemsetAsync(s + n, 0, 12, sc.stream));
        HIP_CHECK(hipMemcpyAsync(s, symbols, (size_t)n * 4, hipMemcpyHostToDevice, sc.stream));
        const int levels = dc3_suffix_array(sc.ctx, s, (u32)n, sigma, sa);
        HIP_CHECK(hipMemcpyAsync(sa_out, sa, (size_t)n * 4, hipMemcpyDeviceToHost, sc.stream));
        HIP_CHECK(hipStreamSynchronize(sc.stream));
        if (levels_out) *levels_out = levels;
    });
}

}  // extern "C"

// ---- several devices in one process ------------------------------------------------------------------
#include "multi.h"
#include "format.h"
