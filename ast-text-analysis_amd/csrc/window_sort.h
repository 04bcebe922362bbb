// window_sort.h -- the window sort over all suffixes (construction A of dc3.h's header) and the
// level-0 machinery it shares with DC3's sample sort on the byte stream:
//
//   WindowSrc / KeyNeqWindowIn   w-symbol window keys (generated inside the first radix pass) and the
//                                naming predicate on the sorted keys
//   lvl0_place_kernel            one pass that places every untied suffix and every small tie group,
//                                with the LCP entries (one document)
//   refinement rounds            large tie groups: compacted domains keyed by (group, next window),
//                                prefix doubling for long repeats, direct endgame on small domains
//   dc3_level0_bytes             the host driver of all of the above (all suffixes: n0 == 0; DC3 sample: n0 > 0)
//   window_suffix_sort           construction A
#pragma once
#include "common.h"
#include "radix_sort.h"
#include "scan.h"
#include "ht_code.h"
#include <math.h>
#include <algorithm>

__device__ __forceinline__ u32 dc3_sample_pos(u32 t, u32 n0)
{
    return t < n0 ? 3u * t + 1u : 3u * (t - n0) + 2u;
}

// Level 0 on the byte stream runs either over the DC3 sample (element t = sample number, n0 > 0)
// or over ALL suffixes (n0 == 0: element t = text position), see window_suffix_sort.
__device__ __forceinline__ u32 lvl0_pos(u32 t, u32 n0) { return n0 ? dc3_sample_pos(t, n0) : t; }
#include "lds_group_sort.h"
#include "persist_rounds.h"
// (the test knobs of this file live in Ctx::knobs -- common.h: Knobs)
static const bool g_trace = getenv("EAST_HIP_TRACE") != nullptr;   // per-round progress on stderr
struct FusedAbort {};           // the fused finish met a repeat too long to order directly: the level is redone with the full sort

#define RESOLVE_MAX_LEN 2048            // longest direct comparison of two suffixes (symbols)

// ---- window keys -------------------------------------------------------------------------------
// A suffix is keyed by its first w symbols (w >= 3, w*bits <= 64), packed at `bits` per symbol.
// Level 0 of an EASA build: every symbol >= the terminator class is a unique string terminator,
// ordered by its position in the corpus.  All terminators share the code term_first in the key
// and the symbols behind the first terminator are dropped; suffixes are enumerated in text order,
// so the STABLE sort leaves equal keys in position order = terminator order, and a key holding a
// terminator is unique.  For DC3's sample (n0 > 0) any order-preserving name over a window that
// covers the triple keeps DC3 correct -- comparing (name(i), name(i+3), ...) still walks the two
// suffixes left to right.
// The key bits left over below the w full symbols (`spare`) take the top bits of symbol w+1:
// still order-preserving, and it thins the ties out further for free.
// Several documents in one shard (all-suffix mode): the document number sits ABOVE the window in the key,
// so one sort yields every document's own suffix array side by side -- no partition afterwards --
// and the window only has to tell the suffixes of ONE document apart.
struct DocKey {
    const u32 *doc_off = nullptr;   // n_docs + 1 offsets into the shard (device)
    u32 n_docs = 1;
    int bits = 0;                   // bit_width(n_docs - 1); 0 = one document
    const u32 *tile_doc = nullptr;  // document of position t << DOC_TILE_SHIFT (filled by dc3_level0_bytes)
    const u32 *h_doc_off = nullptr; // the offsets on the host (nullptr in a sizing run)
    // Segmented sort (radix_sort.h: RsSeg; set by window_suffix_sort): every pass keeps the documents in their own ranges,
    // the keys hold NO document number (bits = 0 then) -- 6-8 more bits of text in the same passes.  What told the
    // documents apart in the keys still holds at the seams: a document's last ranks are its terminator-first suffixes
    // (the terminator class is the largest code), and a key that holds a terminator never ties.
    RsSeg seg;
};
#define DOC_TILE_SHIFT 12

// The score walk's k-gram bucket tables (score.h) can be read off the sorted window keys for free: rank j
// opens a bucket of its document iff the leading k symbol fields (or the document number) of its key
// differ from the key before -- exactly "k-gram class code differs", terminator class and the zeroes
// behind it included.  The placement pass writes those entries (kg[d][code] = rank inside the document;
// the table was preset to 0xFFFFFFFF), the score side only has to run the suffix-minimum fill.
// pairs (k >= 2): the table of the last level holds 8-byte entries {rank, text position of the suffix at that rank} and
// stays UNFILLED (empty buckets keep 0xFFFFFFFF; the walk of score.h looks for a bucket's end among the next entries),
// and the levels above it get a table of their own, kg3 (k - 1 symbols, 4-byte entries, filled on the score side: it is
// A times smaller).  A walk that arrives in a bucket of ONE suffix -- most do -- then has that suffix's position with
// the entry it read anyway, instead of behind a dependent read of the suffix array; the position is only ever used for
// such buckets, whose suffix is untied and therefore final when the mark is written.
struct KgMark {
    u32 *kg = nullptr;              // n_docs rows of bins + 1 entries
    int k = 0;                      // on entry: the largest k the table has room for; on return: the k marked (<= w)
    u32 A = 0, bins = 0;            // alphabet of the class codes (sigma_text + 2), A^k
    const u32 *doc_off = nullptr;
    u32 n_docs = 1;
    u32 *kg3 = nullptr;             // pairs: n_docs rows of bins / A + 1 entries
    int pairs = 0;
    int by_rank = 0;                // segmented sort: the keys hold no document number -- the document comes from the rank
};

// the document of rank j (documents side by side: d with doc_off[d] <= j < doc_off[d + 1])
__device__ __forceinline__ u32 doc_of_rank(const u32 *__restrict__ doc_off, u32 n_docs, u32 j)
{
    u32 lo = 0, hi = n_docs;
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (doc_off[mid] <= j) lo = mid; else hi = mid;
    }
    return lo;
}

// Segmented first-level sort: the sorted keys are sorted inside every document's range of ranks only -- "an equal key
// `limit` places away means more than `limit` equal keys around this one" holds inside a document, and the kernels that
// use it ask for the bounds of the document a rank lies in (d0: the document of a rank at or in front of j, found once
// per workgroup).  n_docs = 0: one range, [0, m).
struct SegRanks {
    const u32 *doc_off = nullptr;
    u32 n_docs = 0;
};
// (the document of the stretch's first rank and its range of ranks, staged in LDS by the workgroup: nearly every rank of
// the stretch lies in it, and the question costs two LDS reads instead of two global loads)
struct DocHint {
    u32 d0, lo, hi;
};
__device__ __forceinline__ void seg_doc_bounds(const SegRanks &sr, const DocHint &h, u32 j, u32 &lo, u32 &hi)
{
    if (j < h.hi) { lo = h.lo; hi = h.hi; return; }
    u32 d = h.d0 + 1u;
    while (d + 1 < sr.n_docs && j >= sr.doc_off[d + 1]) d++;
    lo = sr.doc_off[d];
    hi = sr.doc_off[d + 1];
}
__device__ __forceinline__ DocHint doc_hint_of_rank(const u32 *__restrict__ doc_off, u32 n_docs, u32 j)
{
    const u32 d = doc_of_rank(doc_off, n_docs, j);
    return DocHint{d, doc_off[d], doc_off[d + 1]};
}

// rank j (key k, the key before it kp, its suffix v) opens a k-gram bucket of its document?  Then write the mark(s).
// (hint, by_rank only: the document of the first rank of the workgroup's stretch)
// (SEG: 0 / 1 = km.by_rank known at compile time, 2 = read at run time)
template <class K, int SEG = 2>
__device__ __forceinline__ void kg_mark_key(const KgMark &km, int w, int b, int spare, u32 j, K k, K kp, u32 v,
                                            const DocHint &hint = DocHint{0u, 0u, 0u})
{
    const int top = spare + (w - km.k) * b;
    const bool same = j != 0 && (K)(k >> top) == (K)(kp >> top);
    u32 d = 0, dlo = 0;
    bool by_rank = true;
    if (SEG == 1 || (SEG == 2 && km.by_rank)) {
        // equal class codes: the same bucket -- unless this is the first rank of a document; the key before it is then the
        // last of the document before, terminator first, and so is this one
        if (same && ((u32)(k >> (spare + (w - 1) * b)) & ((1u << b) - 1u)) != km.A - 1u) return;
        d = hint.d0;
        dlo = hint.lo;
        if (j >= hint.hi) {                             // (a seam inside the stretch)
            d++;
            while (d + 1 < km.n_docs && j >= km.doc_off[d + 1]) d++;
            dlo = km.doc_off[d];
        }
        if (same && j != dlo) return;
    } else {
        if (same) return;
        d = km.n_docs > 1 ? (u32)(k >> (w * b + spare)) : 0u;   // (one document: no bits above the window)
        by_rank = false;
    }
    u32 code = 0;
    for (int q = 0; q < km.k; q++) code = code * km.A + ((u32)(k >> (spare + (w - 1 - q) * b)) & ((1u << b) - 1u));
    if (code >= km.bins) return;                        // (only in a speculative build that assumed the wrong alphabet)
    // (read here, behind the code: measured -- 0.45 against 0.42 ms for the 64 MiB finish; one document starts at rank 0)
    if (!by_rank) dlo = km.n_docs > 1 ? km.doc_off[d] : 0u;
    const u32 jl = j - dlo;
    if (!km.pairs) { km.kg[(size_t)d * (km.bins + 1) + code] = jl; return; }
    reinterpret_cast<uint2 *>(km.kg)[(size_t)d * (km.bins + 1) + code] = uint2{jl, v};
    const int top3 = top + b;                           // the level above: k - 1 symbols
    if (j == 0 || jl == 0 || (K)(k >> top3) != (K)(kp >> top3)) km.kg3[(size_t)d * (km.bins / km.A + 1) + code / km.A] = jl;
}

__global__ __launch_bounds__(BLOCK) void doc_tiles_kernel(const u32 *__restrict__ doc_off, u32 n_docs, u32 n_tiles,
                                                          u32 *__restrict__ tile_doc)
{
    const u32 t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_tiles) return;
    const u32 p = t << DOC_TILE_SHIFT;
    u32 lo = 0, hi = n_docs;                    // last d with doc_off[d] <= p
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (doc_off[mid] <= p) lo = mid; else hi = mid;
    }
    tile_doc[t] = lo;
}

template <class K> struct WindowSrc {          // the (window key, element) pairs, generated by the first radix pass
    static constexpr int MODE = 1;             // (element by element: DC3's sample, n0 > 0)
    const uint8_t *s8;
    u32 n0;
    int w, b, spare;
    u32 term_first;
    DocKey docs;
    __device__ __forceinline__ K key(u32 u) const
    {
        K key = window(u);
        if (docs.bits) {                            // (all-suffix mode: u is the text position)
            // the document of the 4096-position tile's first position, then a step forward where a
            // document boundary falls inside the tile (two loads, the same for nearly every lane)
            u32 lo = docs.tile_doc[u >> DOC_TILE_SHIFT];
            while (lo + 1 < docs.n_docs && docs.doc_off[lo + 1] <= u) lo++;
            key |= (K)lo << (w * b + spare);
        }
        return key;
    }
    __device__ __forceinline__ K window(u32 u) const
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

// The same keys for ALL suffixes in text order (n0 == 0, element u = text position u), a tile at a time:
// a thread takes 8 consecutive positions, loads their 21 bytes once (three aligned 8-byte loads) and
// rolls the keys out from right to left --
//     key(q) = code(x[q]) on top of [ the fields of key(q + 1) moved down one place ],   all zero behind a terminator
// -- so that a key costs a dozen integer operations instead of its own two unaligned loads and a loop
// over its symbols, and the truncation at the first terminator comes with the recurrence.  The radix
// sort's first pass calls fill_tile (keys into LDS, no round trip through HBM) and hist_tile (digit
// histogram of a tile).
#define TW_RUN 8
template <class K> struct TextWindowGen {
    static constexpr int MODE = 2;
    const uint8_t *s8;                          // byte stream; readable (any content) up to 24 bytes behind the last symbol
    u32 n;
    int w, b, spare;
    u32 term_first;
    DocKey docs;

    __device__ __forceinline__ void prepare() const {}
    // keys of the positions p0 .. p0 + 7 from x = the bytes s8[p0 .. p0 + 24)
    template <int RUN = TW_RUN>
    __device__ __forceinline__ void keys_of_run(const u32 (&x)[6], K (&out)[RUN]) const
    {
        const int top = spare + (w - 1) * b;
        const K fmask = ((K)1 << b) - 1;
        K key = 0;
#pragma unroll
        for (int q = RUN + 12; q >= 0; q--) {
            if (q > RUN + w) continue;          // (uniform: the run starts w + 1 symbols to the right, from zero)
            const u32 c = (x[q >> 2] >> ((q & 3) * 8)) & 0xFFu;
            const bool term = c == 0xFFu;
            const K body = key >> spare;        // the w full fields of key(q + 1)
            const K rest = (K)((body >> b) << spare) | (K)((body & fmask) >> (b - spare));
            key = ((K)(term ? term_first : c) << top) | (term ? (K)0 : rest);
            if (q < RUN) out[q] = key;
        }
    }
    __device__ __forceinline__ void load_run(u32 p0, u32 (&x)[6]) const
    {
        // (p0 is a multiple of 8 where the tiles start at multiples of RS_TILE; a document's own tiles start anywhere)
        uint2 a, c, e;
        __builtin_memcpy(&a, s8 + p0, 8);
        __builtin_memcpy(&c, s8 + p0 + 8, 8);
        __builtin_memcpy(&e, s8 + p0 + 16, 8);
        x[0] = a.x; x[1] = a.y; x[2] = c.x; x[3] = c.y; x[4] = e.x; x[5] = e.y;
    }
    // the document number on top of the keys of a run (several documents per shard)
    __device__ __forceinline__ void add_docs(u32 p0, K (&out)[TW_RUN]) const
    {
        u32 lo = docs.tile_doc[p0 >> DOC_TILE_SHIFT];
        const int at = w * b + spare;
        // (the document of the run's first position; inside the run the next document's offset is compared against, one
        // load per run instead of one per position -- a run of 8 positions crosses a document border once in 10^5 runs)
        while (lo + 1 < docs.n_docs && docs.doc_off[lo + 1] <= p0) lo++;
        u32 next_off = lo + 1 < docs.n_docs ? docs.doc_off[lo + 1] : 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < TW_RUN; q++) {
            while (p0 + q >= next_off) {
                lo++;
                next_off = lo + 1 < docs.n_docs ? docs.doc_off[lo + 1] : 0xFFFFFFFFu;
            }
            out[q] |= (K)lo << at;
        }
    }
    __device__ __forceinline__ void fill_tile(K *s_keys, u32 tile_base, u32 tile_count) const
    {
        if (!docs.bits) {
            // no document number to add: runs of RS_IPT = 4 positions, so that every thread of the scatter workgroup rolls
            // one out (a run of 8 costs 8 + w + 1 steps of the recurrence and leaves half of the threads idle; one of 4
            // costs 4 + w + 1 steps on all of them)
            constexpr int HALF = TW_RUN / 2;
            for (u32 r = threadIdx.x; r < (u32)RS_TILE / HALF; r += RS_THREADS) {
                if (r * HALF >= tile_count) break;
                u32 x[6];
                K k[HALF];
                load_run(tile_base + r * HALF, x);
                keys_of_run<HALF>(x, k);
                if constexpr (sizeof(K) == 4) *reinterpret_cast<uint4 *>(&s_keys[r * HALF]) = uint4{(u32)k[0], (u32)k[1], (u32)k[2], (u32)k[3]};
                else {
#pragma unroll
                    for (int q = 0; q < HALF; q++) s_keys[r * HALF + q] = k[q];
                }
            }
            return;
        }
        for (u32 r = threadIdx.x; r < (u32)RS_TILE / TW_RUN; r += RS_THREADS) {
            if (r * TW_RUN >= tile_count) break;
            u32 x[6];
            K k[TW_RUN];
            load_run(tile_base + r * TW_RUN, x);
            keys_of_run(x, k);
            if (docs.bits) add_docs(tile_base + r * TW_RUN, k);
#pragma unroll
            for (int q = 0; q < TW_RUN; q++) s_keys[r * TW_RUN + q] = k[q];
        }
    }
    // digit histogram of the positions [e0, e1) of a tile, by ONE wave
    template <bool CHECK = true> __device__ __forceinline__ void hist_tile(u32 *mine, u32 e0, u32 e1, int shift, u32 mask) const
    {
        const bool need_docs = docs.bits && w * b + spare < shift + 12;     // (a digit only sees the document number of short windows)
        for (u32 p0 = e0 + lane_id() * TW_RUN; p0 < e1; p0 += WAVE * TW_RUN) {
            u32 x[6];
            K k[TW_RUN];
            load_run(p0, x);
            keys_of_run(x, k);
            if (need_docs) add_docs(p0, k);
#pragma unroll
            for (int q = 0; q < TW_RUN; q++)
                if (p0 + q < e1) radix_hist_add<CHECK>(mine, (u32)(k[q] >> shift) & mask);
        }
    }
};

// The same for keys of VARIABLE-LENGTH code words (ht_code.h): the key of position q is the first `sb` bits of the coded
// suffix -- stream(q) = code(x[q]) on top of stream(q + 1) moved down by its length, nothing behind a terminator's code
// word --, under the document number.  Code words have at least HT_MIN_LEN = 3 bits, so 16 symbols fill the HT_MAX_STREAM =
// 48 stream bits a key may carry: the run of 8 positions is rolled out of its 24 loaded bytes from the right.  enc[] (byte -> code << 8 | length) is staged in
// LDS by prepare(), which the two kernels of the first radix pass call once, workgroup-wide.
__device__ __forceinline__ u32 *ht_enc_lds()
{
    __shared__ u32 t[256];
    return t;
}
static_assert(HT_MAX_STREAM / HT_MIN_LEN <= 16, "HtWindowGen looks 16 symbols ahead: a key must not hold more");
template <class K> struct HtWindowGen {
    static constexpr int MODE = 2;
    const uint8_t *s8;                          // byte stream; readable (any content) up to 24 bytes behind the last symbol
    u32 n;
    int sb;                                     // stream bits of a key (key bits below the document number)
    const u32 *enc;
    DocKey docs;

    __device__ __forceinline__ void prepare() const
    {
        u32 *t = ht_enc_lds();
        for (u32 i = threadIdx.x; i < 256u; i += blockDim.x) t[i] = enc[i];
        __syncthreads();
    }
    __device__ __forceinline__ void keys_of_run(const u32 (&x)[6], K (&out)[TW_RUN]) const
    {
        const u32 *t = ht_enc_lds();
        K st = 0;
#pragma unroll
        for (int q = 23; q >= 0; q--) {
            const u32 c = (x[q >> 2] >> ((q & 3) * 8)) & 0xFFu;
            const u32 e = t[c];
            const int len = (int)(e & 0xFFu);
            const K rest = c == 0xFFu ? (K)0 : (K)(st >> len);
            st = ((K)(e >> 8) << (sb - len)) | rest;
            if (q < TW_RUN) out[q] = st;
        }
    }
    __device__ __forceinline__ void load_run(u32 p0, u32 (&x)[6]) const
    {
        // (p0 is a multiple of 8 where the tiles start at multiples of RS_TILE; a document's own tiles start anywhere)
        uint2 a, c, e;
        __builtin_memcpy(&a, s8 + p0, 8);
        __builtin_memcpy(&c, s8 + p0 + 8, 8);
        __builtin_memcpy(&e, s8 + p0 + 16, 8);
        x[0] = a.x; x[1] = a.y; x[2] = c.x; x[3] = c.y; x[4] = e.x; x[5] = e.y;
    }
    __device__ __forceinline__ void add_docs(u32 p0, K (&out)[TW_RUN]) const
    {
        u32 lo = docs.tile_doc[p0 >> DOC_TILE_SHIFT];
        while (lo + 1 < docs.n_docs && docs.doc_off[lo + 1] <= p0) lo++;
        u32 next_off = lo + 1 < docs.n_docs ? docs.doc_off[lo + 1] : 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < TW_RUN; q++) {
            while (p0 + q >= next_off) {
                lo++;
                next_off = lo + 1 < docs.n_docs ? docs.doc_off[lo + 1] : 0xFFFFFFFFu;
            }
            out[q] |= (K)lo << sb;
        }
    }
    __device__ __forceinline__ void fill_tile(K *s_keys, u32 tile_base, u32 tile_count) const
    {
        for (u32 r = threadIdx.x; r < (u32)RS_TILE / TW_RUN; r += RS_THREADS) {
            if (r * TW_RUN >= tile_count) break;
            u32 x[6];
            K k[TW_RUN];
            load_run(tile_base + r * TW_RUN, x);
            keys_of_run(x, k);
            if (docs.bits) add_docs(tile_base + r * TW_RUN, k);
#pragma unroll
            for (int q = 0; q < TW_RUN; q++) s_keys[r * TW_RUN + q] = k[q];
        }
    }
    template <bool CHECK = true> __device__ __forceinline__ void hist_tile(u32 *mine, u32 e0, u32 e1, int shift, u32 mask) const
    {
        const bool need_docs = docs.bits && sb < shift + 12;
        for (u32 p0 = e0 + lane_id() * TW_RUN; p0 < e1; p0 += WAVE * TW_RUN) {
            u32 x[6];
            K k[TW_RUN];
            load_run(p0, x);
            keys_of_run(x, k);
            if (need_docs) add_docs(p0, k);
#pragma unroll
            for (int q = 0; q < TW_RUN; q++)
                if (p0 + q < e1) radix_hist_add<CHECK>(mine, (u32)(k[q] >> shift) & mask);
        }
    }
};

// Layout of a window key of `used` = w * bits + document bits: whole radix passes are paid for anyway, so
// the key is filled up to a whole number of digits (of the width the sort will pick) with the top bits of
// one more symbol.  Returns the spare bits taken from symbol w + 1.
static int lvl0_spare_bits(int used, int key_bits, int bt, int w)
{
    const int total = std::min(radix_pass_count(used) * RS_DB, key_bits);
    return w < 12 ? std::min(total - used, bt - 1) : 0;
}

// ht_sb != 0: the keys hold variable-length code words (HtWindowGen): ht_sb stream bits under the document number,
// ht_dec = the decode table (global memory; kernels that read many keys stage it in LDS), ht_wmin = the whole symbols
// every key without a terminator holds at least.
template <class K> struct KeyNeqWindowIn {
    static constexpr bool HAS_KEYS = true;
    const K *keys;
    K rep_t, ones, highs;       // terminator code / 1 / top bit replicated into every full-symbol field
    int ht_sb = 0, ht_wmin = 0;
    const uint16_t *ht_dec = nullptr;
    // the coded stream of a key, left-aligned (the document number shifted out at the top)
    __device__ __forceinline__ K ht_stream(K k) const
    {
        return ht_sb >= (int)sizeof(K) * 8 ? k : (K)(k << ((int)sizeof(K) * 8 - ht_sb));
    }
    // bits the streams of k and kp have in common (ht_sb if the keys are equal; -1: different documents)
    __device__ __forceinline__ int ht_common_bits(K k, K kp) const
    {
        const K d = k ^ kp;
        if (!d) return ht_sb;
        int hb;
        if constexpr (sizeof(K) == 4) hb = 31 - __clz((u32)d); else hb = 63 - __clzll((u64)d);
        return hb >= ht_sb ? -1 : ht_sb - 1 - hb;
    }
    // what the key k holds; common = the symbols it shares with kp (the LCP of the two suffixes, if the keys differ);
    // tb: see HtScan::top
    template <class Table> __device__ __forceinline__ HtScan ht_read(K k, K kp, const Table &dec, int tb = 0) const
    {
        const int cb = ht_common_bits(k, kp);
        return ht_scan<K>(ht_stream(k), ht_sb, dec, cb < 0 ? 0 : cb, tb);
    }
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
        if (ht_sb) return (i == 0 || k != keys[i - 1] || ht_read(k, k, ht_dec).term) ? 1u : 0u;
        const K x = k ^ rep_t;
        const bool has_term = ((x - ones) & ~x & highs) != 0;
        return (i == 0 || has_term || k != keys[i - 1]) ? 1u : 0u;
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

// ---- step 2c: refinement of tied names (level 0, byte stream) ---------------------------------
// Natural-language text repeats words and phrases, so half of the suffixes can share their
// w-symbol name.  The tied ones are worked off in rounds, each on the compacted list of what is
// still tied (`domain`: elem[] = the suffixes in their current order, flag[] = 1 where a group of
// equal names starts, slot[] = where each sits in the global order):
//   * small groups (<= REFINE_SMALL_GROUP) are ordered directly on the text, for good;
//   * the members of larger groups are keyed by (group, the NEXT window of symbols) and radix
//     sorted -- the groups stay where they are, their members get ordered by the next symbols --
//     and what is still tied afterwards forms the next, smaller domain.
// A few rounds cover a whole 3-word string.  Long repeats (the domain shrinks slowly): in all-suffix
// mode the rounds switch to prefix doubling (further down) and finish in O(log n) rounds; in DC3's
// sample mode they stop when the domain stalls -- the refined names still are valid DC3 names
// (order-preserving over a window that covers the triple) and feed the recursion.
#ifndef REFINE_SMALL_GROUP
#define REFINE_SMALL_GROUP 6
#endif
#define REFINE_MAX_ROUNDS 32
#define REFINE_ENDGAME_DOMAIN 262144     // domains this small: groups up to REFINE_ENDGAME_GROUP are ordered directly,
#define REFINE_ENDGAME_GROUP 128         // with comparisons of at most REFINE_ENDGAME_LEN symbols (longer: give up)
#define REFINE_ENDGAME_LEN 256
#define REFINE_DOUBLING_LEN 64           // direct comparisons of small groups in prefix-doubling rounds once long repeats are known
#define REFINE_SMALL_INPUT 65536          // inputs this small use the endgame limits from the placement pass on

struct BitIn {                                  // one flag per element, 64 to a word (written by wave ballots)
    const u64 *bits;
    __device__ __forceinline__ u32 operator()(u32 i) const { return (u32)(bits[i >> 6] >> (i & 63u)) & 1u; }
};

// The rank of a set bit among the set bits -- where a kept element goes when the domain is compacted -- from
// the number of set bits in front of its word (an exclusive scan over the WORDS, 1/64 of the elements).
struct PopIn {
    const u64 *bits;
    u32 n_words;
    __device__ __forceinline__ u32 operator()(u32 w) const { return w < n_words ? (u32)__popcll(bits[w]) : 0u; }
};
struct BitRank {
    const u64 *bits;
    const u32 *word_prefix;
    __device__ __forceinline__ u32 operator()(u32 i) const
    {
        return word_prefix[i >> 6] + (u32)__popcll(bits[i >> 6] & (((u64)1 << (i & 63u)) - 1ull));
    }
};

struct FlagArrIn {                              // the naming predicate of a compacted domain
    static constexpr bool HAS_KEYS = false;
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
    if (f.ht_sb) {
        const HtScan r = f.ht_read(k, kp, f.ht_dec);
        whole = k == kp && !r.term;
        return r.common;
    }
    // (the field of a bit position without a division per rank: x / b = (x * ceil(2^16 / b)) >> 16 for the positions of a
    // key, x < 64; the reciprocal is the same for every rank -- uniform, computed once)
    const u32 inv_b = (65536u + (u32)b - 1u) / (u32)b;
    const u64 d = (u64)(k ^ kp);
    u32 mism = (u32)w;                                   // leading symbol fields in common
    if (d) {
        const int hb = 63 - __builtin_clzll(d);
        if (hb >= spare + w * b) mism = 0;              // (different documents: the entry is reset by lcp_doc_starts)
        else if (hb >= spare) mism = (u32)w - 1u - (((u32)(hb - spare) * inv_b) >> 16);
    }
    const K x = k ^ f.rep_t;
    const u64 tz = (u64)((K)(x - f.ones) & ~x & f.highs);    // at most one field holds the terminator code
    const u32 term = tz ? (u32)w - 1u - (((u32)(__builtin_ctzll(tz) - spare) * inv_b) >> 16) : (u32)w;
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
// A comparison longer than max_len (a long repeat inside a SMALL group: duplicated passages) cannot be
// decided here.  The optimistic pass (mode 0) only raises `fail`; the host then restores the domain
// and repeats the pass in two steps, so that all members of a group act alike: mode 1 marks the
// groups that hold such a pair (bit of the group's first position in `bad`), mode 2 leaves those
// groups to the rounds -- as if they were large -- and places the others.
struct LongRepeats {
    u32 *bad = nullptr;
    int mode = 0;
};

// The first 8 symbols behind the common depth of the tied elements of a workgroup's stretch, gathered ONCE per
// element into LDS: a member of a group of g then ranks itself with g - 1 LDS reads instead of 2 (g - 1) text gathers
// (only pairs that agree on all 8 symbols go on reading the text).
struct NextSymbols {
    const u64 *k2 = nullptr;        // k2[x - first] for the domain positions first .. first + count - 1 (valid for tied ones)
    u32 first = 0, count = 0;
};

// The keys and elements of a workgroup's stretch of the sorted domain (plus PLACE_HALO to either side), staged in
// LDS by the placement pass: the tied elements look for the bounds of their groups and for the other members
// there instead of through a dozen scattered global loads each (phase 2 of the pass was bound by those).
// Outside the staged range both fall back to the arrays.
#define PLACE_HALO 8
template <class K> struct TileStarts {
    KeyNeqWindowIn<K> f;
    const K *kt;                    // kt[i - first] = keys[i] for first <= i < first + count
    u32 first, count;
    const u32 *tbits = nullptr;     // variable-length keys: bit (i - first + tskip) = "the key at i holds a terminator" (staged range)
    u32 tskip = 0;
    __device__ __forceinline__ K key(u32 i) const { return i - first < count ? kt[i - first] : f.keys[i]; }
    __device__ __forceinline__ u32 operator()(u32 i) const
    {
        if (i == 0) return 1u;
        const K k = key(i);
        if (k != key(i - 1)) return 1u;
        if (f.ht_sb) {
            if (i - first < count) return (tbits[(i - first + tskip) >> 5] >> ((i - first + tskip) & 31u)) & 1u;
            return f.ht_read(k, k, f.ht_dec).term ? 1u : 0u;
        }
        const K x = k ^ f.rep_t;
        return ((x - f.ones) & ~x & f.highs) != 0 ? 1u : 0u;
    }
};
struct TileElems {
    const u32 *g, *t;               // t[i - first] = g[i] for first <= i < first + count
    u32 first, count;
    __device__ __forceinline__ u32 operator[](u32 i) const { return i - first < count ? t[i - first] : g[i]; }
};

template <class Starts, class LcpFirst, class Elem = const u32 *>
__device__ __forceinline__ u32 lvl0_place_tied(u32 j, const Elem &elem, const Starts &starts,
                                               const u32 *__restrict__ slot, u32 m, const uint8_t *__restrict__ s8,
                                               u32 n0, u32 depth, u32 *__restrict__ order_g,
                                               u32 *__restrict__ names_g, u32 *__restrict__ lcp_g, LcpFirst lcp_first,
                                               u32 *__restrict__ fail, u32 limit = REFINE_SMALL_GROUP,
                                               u32 max_len = RESOLVE_MAX_LEN, u32 *__restrict__ name_of = nullptr,
                                               LongRepeats lr = LongRepeats(), NextSymbols ns = NextSymbols())
{
    const bool first = slot == nullptr;
    const u32 e = elem[j];
    u32 a = j, bnd = j + 1;
    while (a > 0 && !starts(a) && j - a <= limit) a--;
    while (bnd < m && !starts(bnd) && bnd - j <= limit) bnd++;
    if (bnd - a > limit || (lr.mode == 2 && ((lr.bad[a >> 5] >> (a & 31u)) & 1u))) {
        if (first && lr.mode != 1) { order_g[j] = e; if (names_g) names_g[j] = starts(j); }
        return 1;
    }
    const u32 p = lvl0_pos(e, n0);
    const bool cached = ns.k2 && j - ns.first < ns.count;
    const u64 u0 = cached ? ns.k2[j - ns.first] : 0;
    u32 r = 0, best = 0;                                // best: longest common prefix with a smaller member
    for (u32 x = a; x < bnd; x++) {
        if (x == j) continue;
        const u32 p2 = lvl0_pos(elem[x], n0);
        bool decided = false, less = false;             // less: suffix p2 < suffix p
        u32 h = depth;
        for (; h < depth + max_len && !decided; h += 8) {
            const bool first_step = h == depth && cached && x - ns.first < ns.count;
            const u64 u = first_step ? u0 : load_u64_unaligned(s8 + p + h);
            const u64 v = first_step ? ns.k2[x - ns.first] : load_u64_unaligned(s8 + p2 + h);
            const u64 d = u ^ v, z = ~u;
            const u64 tz = (z - 0x0101010101010101ull) & ~z & 0x8080808080808080ull;
            const u32 mism = d ? (u32)__builtin_ctzll(d) >> 3 : 8u;
            const u32 term = tz ? (u32)__builtin_ctzll(tz) >> 3 : 8u;
            if (term < mism) { less = p2 < p; decided = true; h += term; break; }     // both end in (different) terminators
            if (mism < 8u) { less = ((v >> (8 * mism)) & 0xFFu) < ((u >> (8 * mism)) & 0xFFu); decided = true; h += mism; break; }
        }
        if (!decided) {
            if (lr.mode == 1) atomicOr(&lr.bad[a >> 5], 1u << (a & 31u));
            else atomicOr(fail, 1u);                    // (mode 0: the host restores the domain and repeats the pass)
            return 0;
        }
        if (less) { r++; best = h > best ? h : best; }
    }
    if (lr.mode == 1) return 0;                         // (marking only)
    const u32 at = a + r;                               // final place inside the domain
    const u32 at_g = first ? at : slot[at];
    order_g[at_g] = e;
    if (names_g) names_g[first ? j : slot[j]] = 1;
    if (name_of) name_of[p] = at_g;                     // (prefix doubling: a placed suffix is named by its exact position)
    // (its LCP entry: with a smaller member of the group, or -- the first of a group of the keyed first domain -- from
    // the keys; the first of a group of a later domain keeps the entry it got when its group split off, see
    // dc3_refine_writeback_kernel)
    if (lcp_g) {
        if (r > 0) lcp_g[at_g] = best;
        else if (first) lcp_g[at_g] = lcp_first(at);
    }
    return 0;
}

// four consecutive keys with 16-byte loads (p: 16-byte aligned)
template <class K> __device__ __forceinline__ void fin_load4(const K *p, K (&out)[5])
{
    if constexpr (sizeof(K) == 4) {
        const uint4 q = *reinterpret_cast<const uint4 *>(p);
        out[0] = q.x; out[1] = q.y; out[2] = q.z; out[3] = q.w;
    } else {
        const uint4 q0 = reinterpret_cast<const uint4 *>(p)[0], q1 = reinterpret_cast<const uint4 *>(p)[1];
        out[0] = ((u64)q0.y << 32) | q0.x; out[1] = ((u64)q0.w << 32) | q0.z;
        out[2] = ((u64)q1.y << 32) | q1.x; out[3] = ((u64)q1.w << 32) | q1.z;
    }
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
// HT: the keys hold variable-length code words (KeyNeqWindowIn::ht_sb): the decode table is staged in LDS, every thread
// reads its keys once with it (terminator inside? whole symbols? symbols shared with the key before = the LCP entry),
// and a tied suffix is compared with the other members of its group from the depth its own key holds (`w` is then the
// least number of symbols any key holds).
template <class K, bool ENDGAME_LIMITS, bool OPTIMISTIC, bool HT = false>  // (OPTIMISTIC: the first attempt, LongRepeats mode 0 folded in)
__global__ __launch_bounds__(BLOCK) void lvl0_place_kernel(KeyNeqWindowIn<K> f, const u32 *__restrict__ vals, u32 m,
                                                          const uint8_t *__restrict__ s8, u32 n0, int w, int b,
                                                          int spare, u32 *__restrict__ order_g,
                                                          u32 *__restrict__ names_g, u32 *__restrict__ lcp_g,
                                                          u64 *__restrict__ keep, u32 *__restrict__ block_keep,
                                                          u32 *__restrict__ fail, LongRepeats lr_arg, KgMark km,
                                                          uint8_t *__restrict__ xdep0 = nullptr, SegRanks sr = SegRanks())
{
    const LongRepeats lr = OPTIMISTIC ? LongRepeats() : lr_arg;
    constexpr u32 limit = ENDGAME_LIMITS ? REFINE_ENDGAME_GROUP : REFINE_SMALL_GROUP;
    constexpr u32 max_len = ENDGAME_LIMITS ? REFINE_ENDGAME_LEN : RESOLVE_MAX_LEN;
    __shared__ u32 keep_bits[BLOCK * PLACE_IPT / 32];
    __shared__ u32 work[BLOCK * PLACE_IPT];             // the tied elements of this workgroup's stretch
    __shared__ u64 next8[BLOCK * PLACE_IPT];            // their next 8 symbols (NextSymbols)
    __shared__ __attribute__((aligned(16))) K key_tile[BLOCK * PLACE_IPT + 2 * PLACE_HALO];   // TileStarts
    __shared__ __attribute__((aligned(16))) u32 val_tile[BLOCK * PLACE_IPT + 2 * PLACE_HALO];  // TileElems
    __shared__ u32 n_keep, n_work;
    constexpr int TB_WORDS = (BLOCK * PLACE_IPT + 2 * PLACE_HALO + 31) / 32 + 1;
    __shared__ __attribute__((aligned(16))) uint16_t dec_lds[HT ? HT_DEC_SIZE : 8];  // the decode table
    __shared__ u32 term_bits[HT ? TB_WORDS : 1];        // by staged index: the key holds a terminator
    __shared__ uint8_t dep_tile[HT ? BLOCK * PLACE_IPT : 1];   // whole symbols of the stretch's keys
    __shared__ DocHint kg_hint;                         // (segmented sort) the document of the stretch's first rank
    if (threadIdx.x < BLOCK * PLACE_IPT / 32) keep_bits[threadIdx.x] = 0;
    if (threadIdx.x == 0) { n_keep = 0; n_work = 0; }
    if constexpr (HT) {
        // (the table's 8 KiB with two 16-byte loads per thread: a loop of 2-byte copies was sixteen loads in four round trips)
        static_assert(HT_DEC_SIZE == BLOCK * 16, "two 16-byte loads per thread");
        const uint4 *src = reinterpret_cast<const uint4 *>(f.ht_dec) + threadIdx.x * 2u;
        const uint4 d0 = src[0], d1 = src[1];
        uint4 *dst = reinterpret_cast<uint4 *>(dec_lds) + threadIdx.x * 2u;
        dst[0] = d0; dst[1] = d1;
        if (threadIdx.x < TB_WORDS) term_bits[threadIdx.x] = 0;
    }
    // (every load of the prologue is requested in front of the first barrier: behind it -- as this was written until round 5
    // -- the halo's loads and then the stretch's each waited a round trip of their own, after thread 0's search for the
    // stretch's document, which now runs while they are in flight)
    const u32 j0 = (blockIdx.x * BLOCK + threadIdx.x) * PLACE_IPT;
    u32 my_keep = 0;                                    // suffixes this thread left to the rounds
    // PLACE_HALO entries to either side of the stretch (threads 0 .. 2*PLACE_HALO-1, one each)
    const u32 stretch0 = blockIdx.x * (BLOCK * PLACE_IPT);
    const bool halo_left = threadIdx.x < PLACE_HALO;
    const u32 halo_q = halo_left ? threadIdx.x : threadIdx.x - PLACE_HALO;
    const u64 halo_g = halo_left ? (u64)stretch0 + halo_q - PLACE_HALO : (u64)stretch0 + BLOCK * PLACE_IPT + halo_q;    // (wraps below 0: skipped)
    const bool has_halo = threadIdx.x < 2 * PLACE_HALO && (!halo_left || stretch0 >= PLACE_HALO) && halo_g < (u64)m + 8;
    // (unconditional loads -- a lane with nothing to fetch reads entry 0 --: behind a branch the compiler waits for each load
    // where the branch ends, one round trip per load)
    const u64 halo_at = has_halo ? halo_g : 0u;
    const K halo_k = f.keys[halo_at];
    const u32 halo_v = vals[halo_at];
    K k[PLACE_IPT + 2];
    u32 v[PLACE_IPT];
    u32 work_mask = 0;                                  // bit e: rank j0 + e goes to phase 2
    {
        // keys j0 .. j0+3 and their elements: 16-byte loads (the arrays carry 8 spare entries behind m: whole groups of 4);
        // the keys to either side (j0-1, j0+4) come out of the staged tile behind the barrier -- variable-length keys read
        // them before it (the walk over the five keys), from the array
        const u32 jl = j0 < m ? j0 : 0u;
        K k4[5];
        fin_load4<K>(f.keys + jl, k4);
#pragma unroll
        for (int e = 0; e < PLACE_IPT; e++) k[e + 1] = k4[e];
        const uint4 q = *reinterpret_cast<const uint4 *>(vals + jl);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        if constexpr (HT) {
            k[0] = jl > 0 ? f.keys[jl - 1] : (K)0;
            k[PLACE_IPT + 1] = f.keys[jl + PLACE_IPT];
        }
    }
    if (threadIdx.x == 0)
        kg_hint = sr.n_docs ? doc_hint_of_rank(sr.doc_off, sr.n_docs, blockIdx.x * (BLOCK * PLACE_IPT)) : DocHint{0u, 0u, 0u};
    __syncthreads();
    if (has_halo) {
        const u32 at = halo_left ? halo_q : PLACE_HALO + BLOCK * PLACE_IPT + halo_q;
        key_tile[at] = halo_k;
        val_tile[at] = halo_v;
        if constexpr (HT)
            if (halo_g < m && f.ht_read(halo_k, halo_k, dec_lds).term) atomicOr(&term_bits[at >> 5], 1u << (at & 31u));
    }
    if (j0 < m) {
        const u32 at0 = PLACE_HALO + threadIdx.x * PLACE_IPT;
        if constexpr (sizeof(K) == 4) *reinterpret_cast<uint4 *>(&key_tile[at0]) = uint4{(u32)k[1], (u32)k[2], (u32)k[3], (u32)k[4]};
        else {
#pragma unroll
            for (int e = 0; e < PLACE_IPT; e++) key_tile[at0 + e] = k[e + 1];
        }
        *reinterpret_cast<uint4 *>(&val_tile[at0]) = uint4{v[0], v[1], v[2], v[3]};
    }
    HtScan sc[PLACE_IPT + 1];                           // (HT) what the thread's keys hold; [PLACE_IPT]: the key behind them
    if constexpr (HT) {
        if (j0 < m) {
            u32 tb = 0;
            HtWalk<K> walk;                             // (one walk over the five keys: see HtWalk)
#pragma unroll
            for (int e = 0; e <= PLACE_IPT; e++) {
                int cb = f.ht_common_bits(k[e + 1], k[e]);
                cb = cb < 0 ? 0 : cb;
                if (e > 0) walk.inherit(cb);
                // (the whole key only where the rank may be tied; else the boundaries its own and the next LCP entry need)
                const bool all = k[e + 1] == k[e] || (e < PLACE_IPT && k[e + 2] == k[e + 1]);
                int cbn = e < PLACE_IPT ? f.ht_common_bits(k[e + 2], k[e + 1]) : 0;
                cbn = cbn < 0 ? 0 : cbn;
                walk.extend(f.ht_stream(k[e + 1]), f.ht_sb, dec_lds, all ? f.ht_sb : (cb > cbn ? cb : cbn));
                sc[e] = walk.result(cb, 0);
                if (e < PLACE_IPT) {
                    if (sc[e].term && j0 + e < m) tb |= 1u << e;
                    dep_tile[threadIdx.x * PLACE_IPT + e] = (uint8_t)sc[e].whole;
                }
            }
            // (per rank: what its key holds beyond the least any key does -- the extra depth of its tie group, should the
            // rounds get it; j0 is a multiple of 4 and the array is padded: one 4-byte store)
            if (xdep0) {
                u32 x4 = 0;
#pragma unroll
                for (int e = 0; e < PLACE_IPT; e++) {
                    const u32 x = sc[e].whole > (u32)f.ht_wmin ? sc[e].whole - (u32)f.ht_wmin : 0u;
                    x4 |= (x > 255u ? 255u : x) << (8 * e);
                }
                *reinterpret_cast<u32 *>(xdep0 + j0) = x4;
            }
            const u32 at0 = PLACE_HALO + threadIdx.x * PLACE_IPT;
            if (tb) {
                atomicOr(&term_bits[at0 >> 5], tb << (at0 & 31u));
                if ((at0 & 31u) + PLACE_IPT > 32u) atomicOr(&term_bits[(at0 >> 5) + 1], tb >> (32u - (at0 & 31u)));
            }
        }
    }
    __syncthreads();                                    // (the staged keys: the test for large groups reads them `limit` places away)
    const DocHint hint = kg_hint;
    if (j0 < m) {
        if constexpr (!HT) {
            // (the first stretch has no left halo: rank 0 has no key before it)
            k[0] = j0 > 0 ? key_tile[PLACE_HALO + threadIdx.x * PLACE_IPT - 1] : (K)0;
            k[PLACE_IPT + 1] = key_tile[PLACE_HALO + threadIdx.x * PLACE_IPT + PLACE_IPT];
        }
        bool start[PLACE_IPT + 1];
#pragma unroll
        for (int e = 0; e <= PLACE_IPT; e++) {
            bool has_term;
            if constexpr (HT) has_term = sc[e].term;
            else {
                const K x = k[e + 1] ^ f.rep_t;
                has_term = ((x - f.ones) & ~x & f.highs) != 0;
            }
            start[e] = j0 + e == 0 || j0 + e >= m || has_term || k[e + 1] != k[e];
        }
        if (OPTIMISTIC && km.kg) {                      // k-gram bucket starts, read off the keys (see KgMark)
#pragma unroll
            for (int e = 0; e < PLACE_IPT; e++)
                if (j0 + e < m) kg_mark_key<K>(km, w, b, spare, j0 + e, k[e + 1], k[e], v[e], hint);
        }
        // every rank: final (its key differs from both neighbours'), a member of a large group (sorted keys: an equal key
        // `limit` places away means more than `limit` equal keys around it -- natural-language text: half of the suffixes;
        // flagged without the exact bounds), or one for phase 2
        u32 fin_mask = 0, large_mask = 0;
#pragma unroll
        for (int e = 0; e < PLACE_IPT; e++) {
            const u32 j = j0 + e;
            if (j >= m) break;
            if (start[e] && start[e + 1]) fin_mask |= 1u << e;
#ifdef PLACE_DIAG_NOLARGEREAD
            else if (k[e] == k[e + 1] && k[e + 2] == k[e + 1]) large_mask |= 1u << e;
#else
            else if (k[e] == k[e + 1] && k[e + 2] == k[e + 1]) {                   // (tied on both sides: worth two more reads)
                bool big;
                u32 dlo = 0, dhi = m;                                              // (the sorted range the rank lies in)
                if (sr.n_docs) seg_doc_bounds(sr, hint, j, dlo, dhi);
                if constexpr (limit <= PLACE_HALO) {                               // (out of the staged keys)
                    const u32 at = PLACE_HALO + threadIdx.x * PLACE_IPT + e;
                    big = (j >= dlo + limit && key_tile[at - limit] == k[e + 1]) || (j + limit < dhi && key_tile[at + limit] == k[e + 1]);
                } else {
                    big = (j >= dlo + limit && f.keys[j - limit] == k[e + 1]) || (j + limit < dhi && f.keys[j + limit] == k[e + 1]);
                }
                if (big) large_mask |= 1u << e;
            }
#endif
        }
        if (large_mask) {
            const u32 local0 = threadIdx.x * PLACE_IPT;                           // (a thread's four bits lie in one word)
            atomicOr(&keep_bits[local0 >> 5], large_mask << (local0 & 31u));
            my_keep += (u32)__popc(large_mask);
        }
        if ((fin_mask | large_mask) == (1u << PLACE_IPT) - 1u) {
            // 16-byte stores.  (The LCP entry of a member of a large group that is not the group's first rank is whatever
            // the two keys say -- the rounds write every such entry when its group splits or is ordered; the first rank's
            // entry is the right one already.)
            *reinterpret_cast<uint4 *>(order_g + j0) = uint4{v[0], v[1], v[2], v[3]};
            if (names_g)
                *reinterpret_cast<uint4 *>(names_g + j0) = uint4{(fin_mask & 1u) ? 1u : (u32)start[0], (fin_mask & 2u) ? 1u : (u32)start[1],
                                                                 (fin_mask & 4u) ? 1u : (u32)start[2], (fin_mask & 8u) ? 1u : (u32)start[3]};
            if (lcp_g) {
                u32 h[PLACE_IPT];
#pragma unroll
                for (int e = 0; e < PLACE_IPT; e++) {
                    bool whole;
                    if constexpr (HT) h[e] = j0 + e > 0 ? sc[e].common : 0u;
                    else h[e] = j0 + e > 0 ? lvl0_lcp_of_key_pair(f, w, b, spare, k[e + 1], k[e], whole) : 0u;
                }
                *reinterpret_cast<uint4 *>(lcp_g + j0) = uint4{h[0], h[1], h[2], h[3]};
            }
        } else {
#pragma unroll
            for (int e = 0; e < PLACE_IPT; e++) {
                const u32 j = j0 + e;
                if (j >= m) break;
                if ((fin_mask >> e) & 1u) {
                    order_g[j] = v[e];
                    if (names_g) names_g[j] = 1;
                    if (lcp_g) {
                        bool whole;
                        if constexpr (HT) lcp_g[j] = j > 0 ? sc[e].common : 0u;
                        else lcp_g[j] = j > 0 ? lvl0_lcp_of_key_pair(f, w, b, spare, k[e + 1], k[e], whole) : 0u;
                    }
                } else if ((large_mask >> e) & 1u) {
                    order_g[j] = v[e];
                    if (names_g) names_g[j] = start[e];
                } else {
                    work_mask |= 1u << e;                // phase 2
                }
            }
        }
    }
    {   // the work list: one LDS atomic per wavefront (natural-language text sends half of the ranks there)
        const u32 wc = (u32)__popc(work_mask);
        const u32 inc = wave_inclusive_sum(wc);
        u32 wbase = 0;
        if (lane_id() == 63u && inc) wbase = atomicAdd(&n_work, inc);
        wbase = __shfl(wbase, 63, WAVE) + inc - wc;
#pragma unroll
        for (int e = 0; e < PLACE_IPT; e++)
            if ((work_mask >> e) & 1u) work[wbase++] = j0 + e;
    }
    __syncthreads();
    // phase 2: the tied elements, one per thread, so that their text gathers run side by side
    // instead of one after the other inside the thread that met them.  First every one of them fetches the 8
    // symbols behind the window once (NextSymbols), then they rank themselves inside their groups.
#ifdef PLACE_DIAG_NOPHASE2
    const u32 todo = 0;
#else
    const u32 todo = n_work;
#endif
    const u32 stretch = blockIdx.x * (BLOCK * PLACE_IPT);
    for (u32 i = threadIdx.x; i < todo; i += BLOCK) {
        const u32 j = work[i];
        const u32 dep = HT ? (u32)dep_tile[j - stretch] : (u32)w;       // (the members of a group hold the same key)
        next8[j - stretch] = load_u64_unaligned(s8 + lvl0_pos(val_tile[PLACE_HALO + j - stretch], n0) + dep);
    }
    __syncthreads();
    const NextSymbols ns{next8, stretch, (u32)(BLOCK * PLACE_IPT)};
    // the staged range: the stretch and PLACE_HALO to either side, as far as it exists
    const u32 t_first = stretch >= PLACE_HALO ? stretch - PLACE_HALO : stretch;
    const u32 t_skip = stretch >= PLACE_HALO ? 0u : (u32)PLACE_HALO;               // (no left halo in the first stretch)
    const u64 t_end = std::min<u64>((u64)stretch + BLOCK * PLACE_IPT + PLACE_HALO, (u64)m + 8);
    const u32 t_count = (u32)(t_end - t_first);
    const TileStarts<K> tstarts{f, key_tile + t_skip, t_first, t_count, HT ? term_bits : (const u32 *)nullptr, t_skip};
    const TileElems telems{vals, val_tile + t_skip, t_first, t_count};
    for (u32 i = threadIdx.x; i < todo; i += BLOCK) {
        const u32 j = work[i];
        auto lcp_first = [&](u32 at) -> u32 {
            bool whole;
            if constexpr (HT) return at > 0 ? f.ht_read(tstarts.key(at), tstarts.key(at - 1), dec_lds).common : 0u;
            else return at > 0 ? lvl0_lcp_of_key_pair(f, w, b, spare, tstarts.key(at), tstarts.key(at - 1), whole) : 0u;
        };
        const u32 dep = HT ? (u32)dep_tile[j - stretch] : (u32)w;
        if (lvl0_place_tied(j, telems, tstarts, (const u32 *)nullptr, m, s8, n0, dep, order_g, names_g, lcp_g, lcp_first, fail,
                            limit, max_len, (u32 *)nullptr, lr, ns)) {
            const u32 local = j - blockIdx.x * (BLOCK * PLACE_IPT);
            atomicOr(&keep_bits[local >> 5], 1u << (local & 31u));
            my_keep++;
        }
    }
    // (the count: one LDS atomic per wavefront -- natural-language text keeps hundreds of suffixes per stretch, and as
    // many adds to one word queue up behind one another)
    my_keep = wave_sum(my_keep);
    if (lane_id() == 0 && my_keep) atomicAdd(&n_keep, my_keep);
    __syncthreads();
    // (entries m.. are 0: the exclusive scan over m + 1 entries yields the total)
    if (threadIdx.x < BLOCK * PLACE_IPT / 64)
        keep[(size_t)blockIdx.x * (BLOCK * PLACE_IPT / 64) + threadIdx.x] =
            ((u64)keep_bits[2 * threadIdx.x + 1] << 32) | keep_bits[2 * threadIdx.x];
    if (threadIdx.x == 0) block_keep[blockIdx.x] = n_keep;
}

// ---- the fused end of the sort: last digit + placement in one pass ---------------------------------
// The global LSD passes stop above the low `L` key bits: the pairs arrive sorted (stably) by the TOP part of
// their keys -- document number, the leading symbols -- with the suffixes of a *bucket* (equal top part)
// still in text order.  A window wide enough to tell most suffixes apart leaves a handful of suffixes per
// bucket, so the last digit is not worth a trip through HBM: a workgroup stages a stretch of FIN_CHUNK
// pairs (+ FIN_G to either side) in LDS, every member of a bucket of at most FIN_G suffixes finds its rank
// inside the bucket by counting the members with smaller low bits (equal ones: text order, the sort stays
// stable), the pairs are permuted inside LDS, and the placement logic of lvl0_place_kernel runs on the
// staged, now fully sorted keys -- suffix array, LCP entries, k-gram bucket starts, direct ordering of the
// small tie groups.  One global radix pass (20 B per suffix) and the re-read of the sorted pairs are gone.
//   * A bucket belongs to the workgroup in whose stretch it STARTS (it may run up to FIN_G - 1 positions
//     into the next stretch; the neighbour sees from its own halo that the bucket is not its own).
//   * The LCP entry of a bucket's first rank only depends on the top parts of the two keys (they differ
//     there), so any member of the bucket in front serves as "the key before".
//   * A bucket of more than FIN_G suffixes whose top part holds a terminator is constant -- the fields
//     behind a terminator are zero --: equal keys in text order, final as they stand.
//   * Any other bucket of more than FIN_G suffixes (skewed text) is handed to the refinement rounds as
//     ONE tie group: its members agree on the symbols that lie wholly inside the top part, and that
//     number of symbols is the depth the rounds start from (dc3_level0_bytes: depth0).  Large groups of
//     equal FULL keys join them at the same depth (they agree on more, which does no harm).
#define FIN_IPT 4
#define FIN_CHUNK (BLOCK * FIN_IPT)
// FIN_G = the largest bucket ordered here; the kernel exists for 64 (the default) and 32 (a smaller halo: 3 % less to
// stage, for inputs whose buckets are expected to hold at most ten suffixes)
template <int G> struct FinGeom {
    static constexpr int LEFT = G + 4;                  // staged entries in front of the stretch (>= G + 1; a multiple of 4: a thread's four flags share a word)
    static constexpr int RIGHT = G + 4;                 // ... behind it (>= G + 1)
    static constexpr int STAGE = LEFT + FIN_CHUNK + RIGHT;
    static constexpr int WORDS = (STAGE + 31) / 32 + 1;
    static constexpr int SCAN_WORDS = (G + 4 + 31) / 32 + 1;
};

template <class K> struct FinishArgs {
    const K *keys;
    const u32 *vals;
    u32 m;
    const uint8_t *s8;
    int w, b, spare, low_bits;
    u32 inv_b;                              // ceil(2^16 / b): (x * inv_b) >> 16 = x / b for the bit positions of a key
    K rep_t, ones, highs;                   // KeyNeqWindowIn's constants over all w symbol fields
    K top_mask;                             // the bits of the fields that lie wholly inside the top part
    int kg_top;                             // 1: the k-gram class code reaches below the top part (marks inside a handed-over bucket would be missed)
    u32 *order_g, *lcp_g;
    u64 *keep, *gstart;                     // one bit per rank (zeroed; OR-ed into): left to the rounds / first of its group
    u32 *block_keep, *fail, *kg_bad;
    KgMark km;
    // variable-length code words in the keys (ht_code.h; ht_sb = 0: fixed-width fields): stream bits, the least whole
    // symbols the TOP part of a key holds (the depth the rounds start from), the decode table, and per rank handed to the
    // rounds what its group shares beyond that depth
    int ht_sb = 0, ht_wmin = 0;
    const uint16_t *ht_dec = nullptr;
    uint8_t *xdep0 = nullptr;
    // (segmented sort: km.by_rank, with km.doc_off / km.n_docs = the documents' ranges of ranks -- set whether or not
    // k-gram marks are written)
};

// highest set bit of fl[] in [lo, i] / lowest in [i, hi]; -1 if none (the ranges span at most G + 4 bits)
template <class GE> __device__ __forceinline__ int fin_prev_bit(const u32 *fl, int i, int lo)
{
    int found = -1;
    int wi = i >> 5;
    u32 word = fl[wi] & (0xFFFFFFFFu >> (31 - (i & 31)));
#pragma unroll
    for (int step = 0; step < GE::SCAN_WORDS; step++) {
        if (found < 0 && word) found = (wi << 5) + 31 - __clz(word);
        if (found < 0 && wi > 0) { wi--; word = fl[wi]; } else word = 0;
    }
    return found >= lo ? found : -1;
}
template <class GE> __device__ __forceinline__ int fin_next_bit(const u32 *fl, int i, int hi)
{
    int found = -1;
    int wi = i >> 5;
    u32 word = fl[wi] & (0xFFFFFFFFu << (i & 31));
#pragma unroll
    for (int step = 0; step < GE::SCAN_WORDS; step++) {
        if (found < 0 && word) found = (wi << 5) + __ffs(word) - 1;
        if (found < 0 && wi + 1 < GE::WORDS) { wi++; word = fl[wi]; } else word = 0;
    }
    return found >= 0 && found <= hi ? found : -1;
}

// lvl0_lcp_of_key_pair without the divisions by the symbol width
template <class K>
__device__ __forceinline__ u32 fin_lcp_of_key_pair(const FinishArgs<K> &a, K k, K kp)
{
    const int w = a.w, b = a.b, spare = a.spare;
    const K d = k ^ kp;
    u32 mism = (u32)w;
    if (d) {
        int hb;
        if constexpr (sizeof(K) == 4) hb = 31 - __clz((u32)d); else hb = 63 - __clzll((u64)d);
        if (hb >= spare + w * b) mism = 0;              // (different documents: the entry is reset by lcp_doc_starts)
        else if (hb >= spare) mism = (u32)w - 1u - (((u32)(hb - spare) * a.inv_b) >> 16);
    }
    const K x = k ^ a.rep_t;
    const K tz = (K)(x - a.ones) & ~x & a.highs;        // at most one field holds the terminator code
    u32 term = (u32)w;
    if (tz) {
        int lb;
        if constexpr (sizeof(K) == 4) lb = __ffs((u32)tz) - 1; else lb = __ffsll((unsigned long long)tz) - 1;
        term = (u32)w - 1u - (((u32)(lb - spare) * a.inv_b) >> 16);
    }
    return mism < term ? mism : term;
}

// lvl0_place_tied for the fused finish: a member of a group of equal keys ranks itself inside the group by comparing
// the suffixes from offset w on and is placed for good (suffix array, LCP entry).  The group lies inside its bucket,
// so everything touched is staged (staged indices throughout); next4[] holds the 4 symbols behind the window of every
// tied pair (enough to tell nearly all of them apart; longer comparisons go on reading the text, 8 symbols a step).
// Returns 1 when the group has more than `limit` members (left to the rounds).
// (dec8 / depth_i: variable-length keys -- the decode table in LDS and the whole symbols the member's key holds: the depth
// the comparisons start from.  A tied member's key holds no terminator, and the members of its group hold the same key.)
template <class K>
__device__ __forceinline__ u32 fin_place_tied(const FinishArgs<K> &a, int i, const K *kt, const u32 *vt, const u32 *next4,
                                              u32 base, u32 limit, u32 max_len, const uint8_t *dec8 = nullptr, u32 depth_i = 0)
{
    auto starts = [&](int x) -> bool {
        if (base + (u32)x == 0) return true;
        const K k = kt[x], xx = k ^ a.rep_t;
        if (a.ht_sb) return k != kt[x - 1];
        return ((K)(xx - a.ones) & ~xx & a.highs) != 0 || k != kt[x - 1];
    };
    int lo = i, hi = i + 1;
    while (!starts(lo) && (u32)(i - lo) <= limit) lo--;
    while (base + (u32)hi < a.m && !starts(hi) && (u32)(hi - i) <= limit) hi++;
    if ((u32)(hi - lo) > limit) return 1;
    const uint8_t *s8 = a.s8;
    const u32 depth = a.ht_sb ? depth_i : (u32)a.w, p = vt[i];
    const u32 u0 = next4[i];
    u32 r = 0, best = 0;                                // best: longest common prefix with a smaller member
    for (int x = lo; x < hi; x++) {
        if (x == i) continue;
        const u32 p2 = vt[x];
        bool decided = false, less = false;             // less: suffix p2 < suffix p
        u32 h = depth;
        {
            const u32 v0 = next4[x];
            const u32 d = u0 ^ v0, z = ~u0;
            const u32 tz = (z - 0x01010101u) & ~z & 0x80808080u;
            const u32 mism = d ? (u32)(__ffs(d) - 1) >> 3 : 4u;
            const u32 term = tz ? (u32)(__ffs(tz) - 1) >> 3 : 4u;
            if (term < mism) { less = p2 < p; decided = true; h += term; }        // both end in (different) terminators
            else if (mism < 4u) { less = ((v0 >> (8 * mism)) & 0xFFu) < ((u0 >> (8 * mism)) & 0xFFu); decided = true; h += mism; }
            else h += 4;
        }
        for (; h < depth + max_len && !decided; h += 8) {
            const u64 u = load_u64_unaligned(s8 + p + h), v = load_u64_unaligned(s8 + p2 + h);
            const u64 d = u ^ v, z = ~u;
            const u64 tz = (z - 0x0101010101010101ull) & ~z & 0x8080808080808080ull;
            const u32 mism = d ? (u32)__builtin_ctzll(d) >> 3 : 8u;
            const u32 term = tz ? (u32)__builtin_ctzll(tz) >> 3 : 8u;
            if (term < mism) { less = p2 < p; decided = true; h += term; break; }
            if (mism < 8u) { less = ((v >> (8 * mism)) & 0xFFu) < ((u >> (8 * mism)) & 0xFFu); decided = true; h += mism; break; }
        }
        if (!decided) { atomicOr(a.fail, 1u); return 0; }   // (a repeat too long to compare: the host redoes the level with the full sort)
        if (less) { r++; best = h > best ? h : best; }
    }
    const int at = lo + (int)r;
    const u32 at_g = base + (u32)at;
    a.order_g[at_g] = p;
    if (a.lcp_g) {
        u32 h = best;
        if (r == 0) {
            if (at_g == 0) h = 0;
            else if (a.ht_sb) {
                const KeyNeqWindowIn<K> hf{nullptr, 0, 0, 0, a.ht_sb, a.ht_wmin, a.ht_dec};
                h = hf.ht_read(kt[at], kt[at - 1], dec8).common;
            } else h = fin_lcp_of_key_pair(a, kt[at], kt[at - 1]);
        }
        a.lcp_g[at_g] = h;
    }
    return 0;
}

template <class K, bool ENDGAME_LIMITS, int FIN_G, bool HT = false, bool SEG = false>
__global__ __launch_bounds__(BLOCK) void lvl0_finish_kernel(FinishArgs<K> a)
{
    using GE = FinGeom<FIN_G>;
    constexpr int FIN_LEFT = GE::LEFT, FIN_STAGE = GE::STAGE, FIN_WORDS = GE::WORDS;
    constexpr u32 limit = ENDGAME_LIMITS ? (REFINE_ENDGAME_GROUP < FIN_G ? REFINE_ENDGAME_GROUP : FIN_G) : REFINE_SMALL_GROUP;
    constexpr u32 max_len = ENDGAME_LIMITS ? REFINE_ENDGAME_LEN : RESOLVE_MAX_LEN;
    constexpr int HELD = FIN_IPT + 1;                   // a thread holds 4 pairs of the stretch and (threads 0 .. FIN_G) one of the right halo
    constexpr int G = FIN_G;
    __shared__ __attribute__((aligned(16))) K kt[FIN_STAGE];
    __shared__ __attribute__((aligned(16))) u32 vt[FIN_STAGE];
    __shared__ u32 next4[FIN_STAGE];                    // the 4 symbols behind the window of the tied pairs; before: comp
    __shared__ uint16_t work[FIN_CHUNK + FIN_G];
    __shared__ u32 fl[FIN_WORDS];                       // bucket starts, by staged index
    __shared__ u32 keep_bits[FIN_WORDS], gs_bits[FIN_WORDS];
    __shared__ u32 n_keep, n_work;
    __shared__ DocHint kg_hint;                         // (segmented sort) the document of the stretch's first rank
    __shared__ __attribute__((aligned(16))) uint8_t dec8[HT ? HT_DEC_SIZE : 16];   // variable-length keys: the decode table (length, terminator bit)
    __shared__ uint8_t wdep[HT ? FIN_CHUNK + FIN_G : 1];   // ... the whole symbols of the tied members' keys, by work list index
    const KeyNeqWindowIn<K> hf{a.keys, a.rep_t, a.ones, a.highs, a.ht_sb, a.ht_wmin, a.ht_dec};
    u32 *comp = next4;                                  // (bucket start, low key bits, staged index) of every pair; next4 is used after the ranking
    const u32 m = a.m;
    const u32 c0 = blockIdx.x * FIN_CHUNK;
    const u32 base = c0 - FIN_LEFT;                     // global rank of staged entry 0 (wraps in the first stretch: such ranks test as > m)
    const int L = a.low_bits;
    const u32 tid = threadIdx.x;
    if (tid < FIN_WORDS) { fl[tid] = 0; keep_bits[tid] = 0; gs_bits[tid] = 0; }
    if (tid == 0) {
        n_keep = 0; n_work = 0;
        if constexpr (SEG) kg_hint = doc_hint_of_rank(a.km.doc_off, a.km.n_docs, c0);
    }
    // ---- stage the pairs ---------------------------------------------------------------------------
    K key[HELD];
    u32 val[HELD];
    const int i0 = (int)(FIN_LEFT + tid * FIN_IPT);     // staged index of the thread's first pair
    const int ih = (int)(FIN_LEFT + FIN_CHUNK + tid);   // ... of its right-halo pair (threads 0 .. FIN_G)
    const u32 j0 = c0 + tid * FIN_IPT;
    u32 valid = 0;                                      // bit e: the held pair e exists (rank < m)
    {
        if (j0 + FIN_IPT <= m) {                        // (16-byte loads)
            fin_load4<K>(a.keys + j0, key);
            const uint4 q = *reinterpret_cast<const uint4 *>(a.vals + j0);
            val[0] = q.x; val[1] = q.y; val[2] = q.z; val[3] = q.w;
            valid = 15u;
        } else {
#pragma unroll
            for (int e = 0; e < FIN_IPT; e++) {
                const bool ok = j0 + e < m;
                key[e] = ok ? a.keys[j0 + e] : (K)0;
                val[e] = ok ? a.vals[j0 + e] : 0u;
                valid |= ok ? 1u << e : 0u;
            }
        }
        // (every load of the prologue is requested before the first one is used: with the staging of the stretch in between,
        // the halo loads waited for the stretch's and for each other -- three round trips where one does)
        key[FIN_IPT] = 0; val[FIN_IPT] = 0;
        if (tid <= (u32)G) {
            const u32 jr = c0 + FIN_CHUNK + tid;        // right halo
            if (jr < m) { key[FIN_IPT] = a.keys[jr]; val[FIN_IPT] = a.vals[jr]; valid |= 16u; }
        }
        K key_l = 0;
        u32 val_l = 0;
        if (tid < (u32)FIN_LEFT) {                      // left halo (never moved)
            const u32 jl = base + tid;
            if (jl < m) { key_l = a.keys[jl]; val_l = a.vals[jl]; }
        }
        if constexpr (HT) {
            // (the low bytes of the table's 4 096 16-bit entries, a thread's sixteen with two 16-byte loads -- requested with
            // the pairs; a loop of 2-byte loads at the top of the kernel was four round trips in front of them)
            static_assert(HT_DEC_SIZE == BLOCK * 16, "two 16-byte loads per thread");
            const uint4 *src = reinterpret_cast<const uint4 *>(a.ht_dec) + tid * 2u;
            const uint4 d0 = src[0], d1 = src[1];
            auto lows = [](u32 x, u32 y) { return (x & 0xFFu) | ((x >> 8) & 0xFF00u) | ((y & 0xFFu) << 16) | ((y << 8) & 0xFF000000u); };
            *reinterpret_cast<uint4 *>(&dec8[tid * 16u]) = uint4{lows(d0.x, d0.y), lows(d0.z, d0.w), lows(d1.x, d1.y), lows(d1.z, d1.w)};
        }
        if constexpr (sizeof(K) == 4) *reinterpret_cast<uint4 *>(&kt[i0]) = uint4{(u32)key[0], (u32)key[1], (u32)key[2], (u32)key[3]};
        else {
#pragma unroll
            for (int e = 0; e < FIN_IPT; e++) kt[i0 + e] = key[e];
        }
        *reinterpret_cast<uint4 *>(&vt[i0]) = uint4{val[0], val[1], val[2], val[3]};
        if (tid <= (u32)G) { kt[ih] = key[FIN_IPT]; vt[ih] = val[FIN_IPT]; }
        if (tid < (u32)FIN_LEFT) { kt[tid] = key_l; vt[tid] = val_l; }
    }
    __syncthreads();
    // ---- bucket starts: the top part differs from the rank before (rank 0 and rank m count as starts) ----
    u32 myfl = 0;                                       // bit e: the held pair e starts a bucket
    {
        K prev = kt[i0 - 1];
#pragma unroll
        for (int e = 0; e < FIN_IPT; e++) {
            const u32 j = j0 + e;
            if (j <= m && (j == 0 || j == m || (K)(key[e] >> L) != (K)(prev >> L))) myfl |= 1u << e;
            prev = key[e];
        }
        if (myfl) atomicOr(&fl[i0 >> 5], myfl << (i0 & 31));
        if (tid <= (u32)G) {
            const u32 j = base + (u32)ih;
            if (j <= m && (j == m || (K)(key[FIN_IPT] >> L) != (K)(kt[ih - 1] >> L))) { myfl |= 16u; atomicOr(&fl[ih >> 5], 1u << (ih & 31)); }
        }
        if (tid >= 1 && tid < (u32)FIN_LEFT) {
            const u32 j = base + tid;
            if (j <= m && (j == 0 || j == m || (K)(kt[tid] >> L) != (K)(kt[tid - 1] >> L))) atomicOr(&fl[tid >> 5], 1u << (tid & 31));
        }
    }
    __syncthreads();
    // ---- every held pair: its bucket [s, en) ----------------------------------------------------------------
    // own: a bucket of at most FIN_G pairs that starts in this workgroup's stretch (it is ordered here);
    // hot: a pair of the stretch in a larger bucket.  Everything else belongs to a neighbour.
    int s[HELD], en[HELD];
    u32 own = 0, hot = 0;
    {
        int run = (myfl & 1u) ? i0 : fin_prev_bit<GE>(fl, i0, i0 - G + 1);
#pragma unroll
        for (int e = 0; e < FIN_IPT; e++) {
            if (e > 0 && ((myfl >> e) & 1u)) run = i0 + e;
            s[e] = run >= i0 + e - G + 1 ? run : -1;
        }
        int nxt = fin_next_bit<GE>(fl, i0 + FIN_IPT, i0 + FIN_IPT - 1 + G);
#pragma unroll
        for (int e = FIN_IPT - 1; e >= 0; e--) {
            en[e] = nxt;
            if ((myfl >> e) & 1u) nxt = i0 + e;
        }
#pragma unroll
        for (int e = 0; e < FIN_IPT; e++) {
            if (!((valid >> e) & 1u)) continue;
            if (s[e] < 0 || en[e] < 0 || en[e] - s[e] > G) hot |= 1u << e;
            else if (s[e] >= (int)FIN_LEFT) own |= 1u << e;         // (s < FIN_LEFT: the bucket of the workgroup before)
        }
        s[FIN_IPT] = en[FIN_IPT] = -1;
        if (valid & 16u) {
            const int sh = (myfl & 16u) ? ih : fin_prev_bit<GE>(fl, ih, ih - G + 1);
            if (sh >= (int)FIN_LEFT && sh < (int)(FIN_LEFT + FIN_CHUNK)) {
                const int eh = fin_next_bit<GE>(fl, ih + 1, sh + G);
                if (eh >= 0) { s[FIN_IPT] = sh; en[FIN_IPT] = eh; own |= 16u; }
            }
        }
    }
    const u32 lowmask = (1u << L) - 1u;
    u32 ck[HELD];
#pragma unroll
    for (int e = 0; e < HELD; e++) {
        const int i = e < FIN_IPT ? i0 + e : ih;
        ck[e] = ((u32)s[e] << 19) | (((u32)key[e] & lowmask) << 11) | (u32)i;
        if ((own >> e) & 1u) comp[i] = ck[e];
    }
    __syncthreads();
    // ---- ranks: the pairs of the stretch [lo, hi) that cover this thread's buckets, counted once for all four ----
    // (comp orders the pairs by bucket first: dest = lo + the number of pairs in [lo, hi) below this one)
    int dest[HELD];
    {
        int lo = FIN_STAGE, hi = 0;
#pragma unroll
        for (int e = 0; e < FIN_IPT; e++)
            if ((own >> e) & 1u) { lo = s[e] < lo ? s[e] : lo; hi = en[e] > hi ? en[e] : hi; }
        u32 c0_ = 0, c1_ = 0, c2_ = 0, c3_ = 0;
#ifndef FIN_DIAG_NORANK
        int x = lo;
        for (; x + 1 < hi; x += 2) {                    // (two pairs per step: one ds_read2)
            const u32 ca = comp[x], cb = comp[x + 1];
            c0_ += (ca < ck[0] ? 1u : 0u) + (cb < ck[0] ? 1u : 0u);
            c1_ += (ca < ck[1] ? 1u : 0u) + (cb < ck[1] ? 1u : 0u);
            c2_ += (ca < ck[2] ? 1u : 0u) + (cb < ck[2] ? 1u : 0u);
            c3_ += (ca < ck[3] ? 1u : 0u) + (cb < ck[3] ? 1u : 0u);
        }
        if (x < hi) {
            const u32 c = comp[x];
            c0_ += c < ck[0] ? 1u : 0u;
            c1_ += c < ck[1] ? 1u : 0u;
            c2_ += c < ck[2] ? 1u : 0u;
            c3_ += c < ck[3] ? 1u : 0u;
        }
#else
        c0_ = (u32)(i0 - lo); c1_ = c0_ + 1; c2_ = c0_ + 2; c3_ = c0_ + 3;
#endif
        dest[0] = lo + (int)c0_; dest[1] = lo + (int)c1_; dest[2] = lo + (int)c2_; dest[3] = lo + (int)c3_;
        dest[FIN_IPT] = -1;
        if (own & 16u) {
            u32 c = 0;
            for (int x = s[FIN_IPT]; x < en[FIN_IPT]; x++) c += comp[x] < ck[FIN_IPT] ? 1u : 0u;
            dest[FIN_IPT] = s[FIN_IPT] + (int)c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < HELD; e++)
        if ((own >> e) & 1u) { kt[dest[e]] = key[e]; vt[dest[e]] = val[e]; }
    __syncthreads();
    // ---- placement on the staged, sorted keys ----------------------------------------------------------
    const DocHint hint = SEG ? kg_hint : DocHint{0u, 0u, 0u};
    const KeyNeqWindowIn<K> f{a.keys, a.rep_t, a.ones, a.highs};   // (keys / vals: never read -- everything the tie code touches is staged; a null pointer here crashes hipcc 7.2)
    const int w = a.w, b = a.b, spare = a.spare;
    u32 my_keep = 0;                                    // suffixes this thread left to the rounds (summed per wavefront at the end)
    auto kg_mark = [&](u32 j, K k, K kp, u32 v) {       // k-gram bucket starts, read off the keys (see KgMark)
        kg_mark_key<K, SEG ? 1 : 0>(a.km, w, b, spare, j, k, kp, v, hint);
    };
    // rank `base + i` (one of this workgroup's): returns true when it is final here (suffix and LCP entry in sa_o / lcp_o,
    // to be stored by the caller); a rank handed to the rounds or a member of a small tie group is dealt with inside
    // (sc: variable-length keys -- what the key holds, as far as this rank needs it: the symbols shared with the key before
    // always, the rest where the rank is in a hot bucket or tied with a neighbour)
    auto place_one = [&](int i, bool is_hot, bool first, K k, K kp, K kn, u32 v, u32 &sa_o, u32 &lcp_o, const HtScan &sc) -> bool {
        const u32 j = base + (u32)i;
        sa_o = v;
        if constexpr (HT) lcp_o = j > 0 ? sc.common : 0u;
        else lcp_o = j > 0 ? fin_lcp_of_key_pair(a, k, kp) : 0u;
        // (a rank left to the rounds: what its group shares beyond the depth the rounds start from)
        auto note_depth = [&](u32 shared) {
            if constexpr (HT) a.xdep0[j] = (uint8_t)(shared > (u32)a.ht_wmin ? (shared - (u32)a.ht_wmin > 255u ? 255u : shared - (u32)a.ht_wmin) : 0u);
        };
        if (is_hot) {
            if (a.km.kg && first) kg_mark(j, k, kp, v);
            bool constant;
            if constexpr (HT) constant = sc.term_top;
            else {
                const K xt = k ^ (a.rep_t & a.top_mask);
                constant = ((K)(xt - (a.ones & a.top_mask)) & ~xt & (a.highs & a.top_mask)) != 0;
            }
            if (constant) return true;                  // a constant bucket: final as it stands
            note_depth(sc.top);
            // (handed to the rounds: the suffix goes to its rank as it is, with the caller's 16-byte stores; the LCP entry is
            // the right one for the group's first rank, and the rounds write every other one when the group splits)
            atomicOr(&keep_bits[i >> 5], 1u << (i & 31));
            my_keep++;
            if (first) atomicOr(&gs_bits[i >> 5], 1u << (i & 31));
            if (a.kg_top && a.km.kg) *a.kg_bad = 1u;
            return true;
        }
        if (a.km.kg) kg_mark(j, k, kp, v);
        bool st, st_next = j + 1 >= m;
        if constexpr (HT) {
            st = j == 0 || sc.term || k != kp;
            st_next = st_next || kn != k || sc.term;    // (an equal key holds a terminator exactly if this one does)
        } else {
            const K x = k ^ f.rep_t;
            st = j == 0 || ((K)(x - f.ones) & ~x & f.highs) != 0 || k != kp;
            if (!st_next) {
                const K xn = kn ^ f.rep_t;
                st_next = ((K)(xn - f.ones) & ~xn & f.highs) != 0 || kn != k;
            }
        }
        if (st && st_next) return true;
        // tied: a group of equal keys inside this bucket (all of it is staged).  More than `limit` equal keys around
        // this one (an equal key `limit` places away -- equal keys share their bucket, so that place is sorted too):
        // a large group, left to the rounds without the exact bounds.
        bool big = false;
        {
            const int lo = i - (int)limit, hi = i + (int)limit;
            u32 dlo = 0, dhi = m;                       // (the sorted range the rank lies in)
            if constexpr (SEG) seg_doc_bounds(SegRanks{a.km.doc_off, a.km.n_docs}, hint, j, dlo, dhi);
            if (lo >= 1 && j >= dlo + limit) big = kt[lo] == k;
            if (!big && hi <= (int)(FIN_LEFT + FIN_CHUNK + FIN_G) && j + limit < dhi) big = kt[hi] == k;   // (staged up to there)
        }
        if (big) {
            atomicOr(&keep_bits[i >> 5], 1u << (i & 31));
            my_keep++;
            if (st) atomicOr(&gs_bits[i >> 5], 1u << (i & 31));
            note_depth(sc.whole);
            return true;                                // (as above: stored by the caller)
        }
        const u32 wi = atomicAdd(&n_work, 1u);
        work[wi] = (uint16_t)i;
        if constexpr (HT) wdep[wi] = (uint8_t)sc.whole;
        return false;
    };
    {
        u32 sa4[FIN_IPT], lc4[FIN_IPT];
        u32 fin = 0;
        const u32 todo4 = (own | hot) & 15u;
        // the sorted keys of the thread's four ranks and of the two next to them, the suffixes: 16-byte LDS reads
        K kq[FIN_IPT + 2];
        u32 vq[FIN_IPT];
        kq[0] = j0 > 0 ? kt[i0 - 1] : (K)0;
        if constexpr (sizeof(K) == 4) {
            const uint4 q = *reinterpret_cast<const uint4 *>(&kt[i0]);
            kq[1] = q.x; kq[2] = q.y; kq[3] = q.z; kq[4] = q.w;
        } else {
#pragma unroll
            for (int e = 0; e < FIN_IPT; e++) kq[e + 1] = kt[i0 + e];
        }
        kq[FIN_IPT + 1] = kt[i0 + FIN_IPT];
        {
            const uint4 q = *reinterpret_cast<const uint4 *>(&vt[i0]);
            vq[0] = q.x; vq[1] = q.y; vq[2] = q.z; vq[3] = q.w;
        }
#ifndef FIN_DIAG_NOPLACE
        // variable-length keys: the four keys are read in one walk (HtWalk) -- the boundaries inside the bits a key shares
        // with the one before are inherited, the rest is decoded only as far as it is needed
        HtScan sc4[FIN_IPT];
        if constexpr (HT) {
            HtWalk<K> walk;
            const int tb = a.ht_sb - L;
#pragma unroll
            for (int e = 0; e < FIN_IPT; e++) {
                const K k = kq[e + 1];
                int cb = hf.ht_common_bits(k, kq[e]);
                const int cb_next = hf.ht_common_bits(kq[e + 2], k);
                if (e == 0) cb = cb < 0 ? 0 : cb; else { cb = cb < 0 ? 0 : cb; walk.inherit(cb); }
                const bool all = ((hot >> e) & 1u) || k == kq[e] || kq[e + 2] == k;       // a hot bucket, or tied with a neighbour
                const int need = all ? a.ht_sb : (cb > cb_next ? cb : cb_next);
                walk.extend(hf.ht_stream(k), a.ht_sb, dec8, need);
                sc4[e] = walk.result(cb, tb);
            }
        } else {
#pragma unroll
            for (int e = 0; e < FIN_IPT; e++) sc4[e] = HtScan{0u, 0u, 0u, false, false};
        }
#pragma unroll
        for (int e = 0; e < FIN_IPT; e++)
            if (((todo4 >> e) & 1u) &&
                place_one(i0 + e, (hot >> e) & 1u, (myfl >> e) & 1u, kq[e + 1], kq[e], kq[e + 2], vq[e], sa4[e], lc4[e], sc4[e]))
                fin |= 1u << e;
#else
#pragma unroll
        for (int e = 0; e < FIN_IPT; e++) { sa4[e] = vq[e]; lc4[e] = (u32)kq[e + 1] ^ (u32)kq[e]; }
        fin = todo4;
#endif
        if (fin == 15u) {
            *reinterpret_cast<uint4 *>(a.order_g + j0) = uint4{sa4[0], sa4[1], sa4[2], sa4[3]};
            if (a.lcp_g) *reinterpret_cast<uint4 *>(a.lcp_g + j0) = uint4{lc4[0], lc4[1], lc4[2], lc4[3]};
        } else {
#pragma unroll
            for (int e = 0; e < FIN_IPT; e++)
                if ((fin >> e) & 1u) {
                    a.order_g[j0 + e] = sa4[e];
                    if (a.lcp_g) a.lcp_g[j0 + e] = lc4[e];
                }
        }
        if (own & 16u) {                                // the overhang of the last bucket that starts in the stretch
            u32 sa1, lc1;
            HtScan sc1{0u, 0u, 0u, false, false};
            if constexpr (HT) sc1 = hf.ht_read(kt[ih], kt[ih - 1], dec8, a.ht_sb - L);
            if (place_one(ih, false, false, kt[ih], kt[ih - 1], kt[ih + 1], vt[ih], sa1, lc1, sc1)) {
                a.order_g[base + (u32)ih] = sa1;
                if (a.lcp_g) a.lcp_g[base + (u32)ih] = lc1;
            }
        }
    }
    __syncthreads();
    // phase 2: the members of small tie groups, one per thread (as in lvl0_place_kernel)
    const u32 todo = n_work;
    for (u32 q = tid; q < todo; q += BLOCK) {
        const u32 i = work[q];
        u32 x4;
        __builtin_memcpy(&x4, a.s8 + vt[i] + (HT ? (u32)wdep[q] : (u32)w), 4);
        next4[i] = x4;
    }
    __syncthreads();
    for (u32 q = tid; q < todo; q += BLOCK) {
        const int i = (int)work[q];
        if (fin_place_tied(a, i, kt, vt, next4, base, limit, max_len, HT ? dec8 : (const uint8_t *)nullptr, HT ? (u32)wdep[q] : 0u)) {
            const u32 j = base + (u32)i;
            a.order_g[j] = vt[i];
            atomicOr(&keep_bits[i >> 5], 1u << (i & 31));
            my_keep++;
            if constexpr (HT) {
                const u32 shared = (u32)wdep[q];
                a.xdep0[j] = (uint8_t)(shared > (u32)a.ht_wmin ? (shared - (u32)a.ht_wmin > 255u ? 255u : shared - (u32)a.ht_wmin) : 0u);
            }
            const K k = kt[i], x = k ^ a.rep_t;
            const bool has_term = HT ? false : ((K)(x - a.ones) & ~x & a.highs) != 0;       // (a tied key holds none)
            if (j == 0 || has_term || k != kt[i - 1]) {     // the first of its group
                atomicOr(&gs_bits[i >> 5], 1u << (i & 31));
                if (a.lcp_g) {
                    if constexpr (HT) a.lcp_g[j] = j > 0 ? hf.ht_read(k, kt[i - 1], dec8).common : 0u;
                    else a.lcp_g[j] = j > 0 ? fin_lcp_of_key_pair(a, k, kt[i - 1]) : 0u;
                }
            }
        }
    }
    my_keep = wave_sum(my_keep);
    if (lane_id() == 0 && my_keep) atomicAdd(&n_keep, my_keep);
    __syncthreads();
    // the bits of the ranks [c0, c0 + FIN_CHUNK + FIN_G): OR-ed into the global words (the overhang shares its words with
    // the next workgroup); nearly all of them are zero
    if (tid < (FIN_CHUNK + FIN_G + 31) / 32) {
        // staged index FIN_LEFT + 32 t .. + 31  ->  global bit c0 + 32 t ..
        const u32 ib = FIN_LEFT + 32u * tid;
        const u32 sh = ib & 31u, wi = ib >> 5;
        u32 kb = keep_bits[wi] >> sh, gb = gs_bits[wi] >> sh;
        if (sh) { kb |= keep_bits[wi + 1] << (32u - sh); gb |= gs_bits[wi + 1] << (32u - sh); }
        const u32 bit0 = c0 + 32u * tid;
        if (kb) atomicOr(reinterpret_cast<u32 *>(a.keep) + (bit0 >> 5), kb);
        if (gb) atomicOr(reinterpret_cast<u32 *>(a.gstart) + (bit0 >> 5), gb);
    }
    if (tid == 0) a.block_keep[blockIdx.x] = n_keep;
}

// group starts kept as one bit per rank (the fused finish): the naming predicate of the first compaction
struct BitStarts {
    static constexpr bool HAS_KEYS = false;
    const u64 *bits;
    __device__ __forceinline__ u32 operator()(u32 i) const { return (u32)(bits[i >> 6] >> (i & 63u)) & 1u; }
};

// LCP entries by comparing the two suffixes from their first symbol (same cap rule as lcp8_kernel)
// ... of a LIST of ranks: the members of the groups that were still open when the rounds went over to prefix doubling
// (their slots, kept when that round compacted its domain) -- a few per cent of all ranks, no pass over the others
__global__ __launch_bounds__(BLOCK) void lvl0_lcp_text_list_kernel(const uint8_t *__restrict__ s8, const u32 *__restrict__ sa,
                                                                   const u32 *__restrict__ ranks, u32 count,
                                                                   u32 *__restrict__ lcp, u32 *__restrict__ capped,
                                                                   LcpBudget budget, bool hinted = false)
{
    // hinted (the persistent rounds, persist_rounds.h): lcp[r] holds what the two suffixes are known to share -- the depth
    // of the round in which they came apart --, and the comparison starts there
    const u32 t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= count) return;
    const u32 r = ranks[t];
    if (r == 0) { lcp[0] = 0; return; }
    const u32 h0 = hinted ? lcp[r] : 0u;
    // (a pair known to share LCP_SOFT_CAP symbols or more goes straight to the finishing pass, whose comparisons run a
    // wavefront wide: one thread walking 10 000 symbols is 300 dependent round trips -- 0.3 ms that a small build waits for)
    if (hinted) budget.per_slot = 0;                    // (... and none of them goes deeper than that here)
    const u32 h = h0 >= LCP_SOFT_CAP ? (LCP_PARTIAL_BIT | h0) : lcp_bytes_capped(s8, sa[r - 1], sa[r], h0, budget);
    if (h & LCP_PARTIAL_BIT) raise_flag(capped);
    lcp[r] = h;
}

// A later, compacted domain of m elements (slot[] = where each sits in the global order): untied
// elements already have their place (written by the previous round's write-back); small groups are
// placed by lvl0_place_tied; members of large groups are marked in keep[] (one bit each).
__global__ __launch_bounds__(BLOCK) void dc3_refine_classify_kernel(const u32 *__restrict__ elem, FlagArrIn starts,
                                                                    const u32 *__restrict__ slot, u32 m,
                                                                    const uint8_t *__restrict__ s8, u32 n0, u32 depth,
                                                                    u32 *__restrict__ order_g, u32 *__restrict__ names_g,
                                                                    u64 *__restrict__ keep, u32 *__restrict__ fail,
                                                                    u32 limit, u32 max_len, u32 *__restrict__ name_of,
                                                                    LongRepeats lr, u32 *__restrict__ lcp_g,
                                                                    const uint2 *__restrict__ only_rest = nullptr,
                                                                    const uint8_t *__restrict__ xdep = nullptr)
{
    // (as in the placement pass: every tied element of the workgroup's stretch fetches the 8 symbols behind the common
    // depth once, and the members of a group rank themselves against those instead of gathering each other's text)
    // only_rest != nullptr: the in-LDS round classified its own tiles (lds_group_sort.h: LgClassify) -- only the positions
    // the global sort took are looked at, and the keep bits are OR-ed into words the host has zeroed
    __shared__ u64 next8[BLOCK];
    const u32 j = blockIdx.x * BLOCK + threadIdx.x;
    u32 my_keep = 0;
    const bool mine = !only_rest || LgUncovered{only_rest, m}(j);       // (a group lies wholly inside or outside the tiles)
    const bool tied = mine && j < m && ((j > 0 && !starts(j)) || (j + 1 < m && !starts(j + 1)));
    if (tied && xdep) depth += (u32)xdep[j];            // (the members of a group share it)
    // (limit 1: nothing is compared -- deep doubling rounds --, so nothing is fetched: the gather was 0.5 ms of a 0.73 ms launch
    // over 16 M elements)
    if (tied && limit > 1u) next8[threadIdx.x] = load_u64_unaligned(s8 + lvl0_pos(elem[j], n0) + depth);
    __syncthreads();
    if (tied)
        my_keep = lvl0_place_tied(j, elem, starts, slot, m, s8, n0, depth, order_g, names_g, lcp_g, NoLcp(), fail, limit, max_len,
                                  name_of, lr, NextSymbols{next8, blockIdx.x * BLOCK, (u32)BLOCK});
    // keep[]: one bit per element (entries m.. of the last word are 0: the exclusive scan over m + 1 yields the total)
    const u64 bal = __ballot(my_keep != 0);
    if (lane_id() == 0) {
        if (!only_rest) keep[j >> 6] = bal;
        else if (bal) atomicOr((unsigned long long *)&keep[j >> 6], (unsigned long long)bal);
    }
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

// the members of large groups, compacted: the radix round's input.  Four consecutive domain positions per thread: their
// keep bits come out of one word, and a thread none of whose positions is kept (half of them, in natural-language text)
// is done after a single load.
#define COMPACT_IPT 4
template <class Starts>
__global__ __launch_bounds__(BLOCK) void dc3_refine_compact_kernel(const u32 *__restrict__ elem, Starts starts,
                                                                   const u32 *__restrict__ slot, BitIn keep,
                                                                   BitRank idx, u32 m,
                                                                   u32 *__restrict__ slot_out, u32 *__restrict__ elem_out,
                                                                   u32 *__restrict__ group_start,
                                                                   u32 *__restrict__ lcp_g = nullptr, int w = 0, int b = 0,
                                                                   int spare = 0, const uint8_t *__restrict__ xdep_in = nullptr,
                                                                   uint8_t *__restrict__ xdep_out = nullptr)
{
    // xdep (variable-length first-level keys): what the group of a position shares BEYOND the rounds' common depth --
    // first domain: written per rank by the placement pass (the whole symbols of the key less the least any key holds);
    // later domains: copied along
    const u32 j0 = (blockIdx.x * BLOCK + threadIdx.x) * COMPACT_IPT;
    if (j0 >= m) return;
    const u64 word = keep.bits[j0 >> 6];
    const u32 nib = (u32)(word >> (j0 & 63u)) & ((1u << COMPACT_IPT) - 1u);
    if (!nib) return;
    // (the rank of the first kept position among the kept ones; the others follow)
    u32 k = idx.word_prefix[j0 >> 6] + (u32)__popcll(word & (((u64)1 << (j0 & 63u)) - 1ull));
    // (what the four positions need is read for all four before any of it is used: one wait instead of one per kept
    // position -- the lines are the same ones)
    u32 st4[COMPACT_IPT], el4[COMPACT_IPT], sl4[COMPACT_IPT];
#pragma unroll
    for (int e = 0; e < COMPACT_IPT; e++) {
        const u32 j = j0 + e < m ? j0 + e : j0;
        st4[e] = starts(j);
        el4[e] = elem[j];
        sl4[e] = slot ? slot[j] : j;
    }
#pragma unroll
    for (int e = 0; e < COMPACT_IPT; e++) {
        const u32 j = j0 + e;
        if (j >= m || !((nib >> e) & 1u)) continue;
        const u32 st = st4[e];
        slot_out[k] = sl4[e];
        elem_out[k] = el4[e];
        group_start[k] = st;
        if (xdep_out) xdep_out[k] = xdep_in ? xdep_in[j] : (uint8_t)0;
        if constexpr (Starts::HAS_KEYS) {
            // the keyed first domain: the first rank of a group left to the rounds gets its LCP entry here, from the
            // two keys -- whichever member ends up there
            if (lcp_g && st) {
                bool whole;
                lcp_g[j] = j > 0 ? lvl0_lcp_of_keys(starts, w, b, spare, j, whole) : 0u;
            }
        }
        k++;
    }
}

// key = (dense group number << w2*b) | next window of w2 symbols at offset `depth`
__global__ __launch_bounds__(BLOCK) void dc3_refine_keys_kernel(const uint8_t *__restrict__ s8,
                                                                const u32 *__restrict__ elems,
                                                                const u32 *__restrict__ group, u32 n_tied, u32 n0,
                                                                u32 depth, int w2, int b, u32 term_first,
                                                                u64 *__restrict__ keys, u32 *__restrict__ vals,
                                                                const uint8_t *__restrict__ xdep = nullptr,
                                                                const u32 *__restrict__ full_idx = nullptr)
{
    const u32 j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n_tied) return;
    const u32 p = lvl0_pos(elems[j], n0) + depth + (xdep ? (u32)xdep[full_idx ? full_idx[j] : j] : 0u);
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
    vals[j] = elems[j];                 // (the suffix itself travels with its key: no look-up after the sort)
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
                                                                const u32 *__restrict__ group, u32 m, u32 depth, int name_bits,
                                                                u64 *__restrict__ keys, u32 *__restrict__ vals)
{
    const u32 j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m) return;
    keys[j] = ((u64)group[j] << name_bits) | (u64)name_of[elems[j] + depth];      // (a name is a rank: bit_width(n) bits, not 32 -- a pass less)
    vals[j] = elems[j];                 // (the suffix itself travels with its key: no look-up after the sort)
}

// after the sort: the slots of the domain receive their members in refined order (globally and as
// the next domain's elem[]), and the naming predicate is updated
__global__ __launch_bounds__(BLOCK) void dc3_refine_writeback_kernel(const u64 *__restrict__ keys,
                                                                     const u32 *__restrict__ vals,
                                                                     const u32 *__restrict__ slots,
                                                                     const u32 *__restrict__ elems, u32 n_tied,
                                                                     u64 rep_t, u64 ones, u64 highs,
                                                                     u32 *__restrict__ order_g, u32 *__restrict__ names_g,
                                                                     u32 *__restrict__ elem_out, u32 *__restrict__ flag_out,
                                                                     u32 *__restrict__ lcp_g = nullptr, u32 depth = 0,
                                                                     int w2 = 0, int b = 0,
                                                                     const u32 *__restrict__ full_idx = nullptr,
                                                                     const uint8_t *__restrict__ xdep = nullptr)
{
    const u32 r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n_tied) return;
    const u64 k = keys[r], kp = r ? keys[r - 1] : 0;
    const u64 x = k ^ rep_t;
    const u64 tz = (x - ones) & ~x & highs;
    const bool has_term = tz != 0;                              // a terminator inside the window: unique
    // (full_idx: the sorted pairs are the compacted rest of a domain; groups keep their stretches, so the r-th of
    // them belongs where the r-th of the rest came from)
    const u32 at = full_idx ? full_idx[r] : r;
    const u32 slot = slots[at], e = vals[r];
    const u32 f = (r == 0 || has_term || k != kp) ? 1u : 0u;
    order_g[slot] = e;
    if (names_g) names_g[slot] = f;
    elem_out[at] = e;
    flag_out[at] = f;
    // symbol windows: where a group splits, the LCP entry of the rank that starts the new group is what the two
    // windows have in common behind the `depth` symbols the group shares (cut at a terminator: two equal terminator
    // codes are different terminators) -- final whichever members end up at the seam.  The first rank of the old
    // group keeps the entry it has.
    const int gshift = w2 * b;
    if (lcp_g && f && r > 0 && (k >> gshift) == (kp >> gshift)) {
        const u64 d = (k ^ kp) & (((u64)1 << gshift) - 1u);
        const u32 inv_b = (65536u + (u32)b - 1u) / (u32)b;   // (uniform: x / b = (x * inv_b) >> 16 for bit positions below 64)
        const u32 mism = d ? (u32)w2 - 1u - (((u32)(63 - __builtin_clzll(d)) * inv_b) >> 16) : (u32)w2;
        const u32 term = tz ? (u32)w2 - 1u - (((u32)__builtin_ctzll(tz) * inv_b) >> 16) : (u32)w2;
        lcp_g[slot] = depth + (xdep ? (u32)xdep[at] : 0u) + (mism < term ? mism : term);
    }
}

// one bit per rank: the slots of the kept members of a domain
__global__ __launch_bounds__(BLOCK) void lvl0_mark_slots_kernel(const u32 *__restrict__ slot, BitIn keep, u32 m,
                                                                u32 *__restrict__ mask)
{
    const u32 j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m || !keep(j)) return;
    const u32 s = slot ? slot[j] : j;
    atomicOr(&mask[s >> 5], 1u << (s & 31u));
}

// what the placement pass left undone, next to the other flags (speculative build: read at the very end):
// the sum of its workgroups' counts (out[] was zeroed with the build's flag words; a few workgroups add into it)
#define SPEC_COUNT_TILE (BLOCK * 16)
__global__ __launch_bounds__(BLOCK) void spec_counts_kernel(const u32 *__restrict__ block_keep, u32 gp,
                                                            const u32 *__restrict__ fail, u32 *__restrict__ out)
{
    __shared__ u32 lds[WAVES_PER_BLOCK];
    const u32 base = blockIdx.x * SPEC_COUNT_TILE;
    u32 sum = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const u32 i = base + j * BLOCK + threadIdx.x;
        if (i < gp) sum += block_keep[i];
    }
    sum = wave_sum(sum);
    if (lane_id() == 0) lds[wave_id()] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 t = lds[0] + lds[1] + lds[2] + lds[3];
        if (t) atomicAdd(&out[0], t);
        if (blockIdx.x == 0) out[1] = *fail;
    }
}

// The persistent rounds (persist_rounds.h): how many of its workgroups the device holds at once -- the launch never asks
// for more -- and the lock that keeps two such launches of one process from sharing the chip half resident each.
#define PERSIST_SMALL_INPUT ((u32)4 << 20)      // inputs whose names cost next to nothing to initialise (0.1 ms)
static std::mutex g_persist_mutex;
struct PersistCap { u32 one_per_cu = 0, large = 0; };      // workgroups resident at once: the resident form / the large form (0: not available)
static PersistCap persist_capacity()
{
    static std::mutex mu;
    static PersistCap cap[64];
    static bool asked[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PersistCap();
    std::lock_guard<std::mutex> lock(mu);
    if (!asked[dev]) {
        asked[dev] = true;
        int per1 = 0, per3 = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per1, refine_persist_kernel<4>, LG_THREADS, 0) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per3, refine_persist2_kernel, LG_THREADS, 0) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); per1 = per3 = 0; }
        // (one workgroup of 1 024 threads per CU: see the launch below)
        if (cus > 0 && per1 >= 1) cap[dev].one_per_cu = (u32)cus;
        if (cus > 0 && per3 >= 1) cap[dev].large = (u32)cus;
    }
    return cap[dev];
}

// Level 0 on the byte stream: the window keys are sorted, then one classify pass places everything
// that is untied or tied in a small group (ordered directly on the text); large groups go through
// refinement rounds (step 2c).  Returns true when sa12 is final (no name string is ever written);
// otherwise the refined names are scanned and scattered into s12 for the recursion (sample mode),
// or the caller falls back to DC3 (all-suffix mode, s12 == nullptr).
#define FIN_LOW_BITS 8                   // key bits the fused finish orders in LDS (one global radix pass less)
#define FIN_MAX_EXPECTED 32.0            // ... when a bucket of the top part is expected to hold at most this many suffixes
// ht != nullptr (all-suffix mode, 32-bit keys): the first-level keys are the first bits of the suffixes coded with
// variable-length code words (ht_code.h); w = the least number of whole symbols a key holds -- the depth the rounds start
// from, every position carrying what its group shares beyond it (xdep).
struct HtKeys {
    const u32 *enc = nullptr;           // device: byte -> code << 8 | length
    const uint16_t *dec = nullptr;      // device: 12 stream bits -> symbol << 8 | terminator << 7 | length
    int max_len = 0;                    // the longest code word
    int sb = 0;                         // stream bits of a key (<= HT_MAX_STREAM), under the document number
};
template <class K>
static bool dc3_level0_bytes(Ctx &ctx, const uint8_t *s8, u32 n0, u32 n02, int w, int bt, u32 term_first, u32 *sa12,
                             u32 *s12, u32 &n_names, u32 *lcp_out = nullptr, u32 *lcp_capped = nullptr,
                             DocKey docs = DocKey(), KgMark *kg_mark = nullptr, bool allow_fused = false, u32 longest = 0,
                             const HtKeys *ht = nullptr)
{
    Arena &ar = *ctx.arena;
    const u32 g02 = ceil_div_u32((u64)n02 + 1, BLOCK);
    const int ht_sb = ht ? ht->sb : 0;
    // whole passes are paid for anyway: fill the last digit with the top bits of the next symbol
    // (the document number, if any, sits above window and spare bits)
    const int spare = ht ? 0 : lvl0_spare_bits(w * bt + docs.bits, (int)sizeof(K) * 8, bt, w);
    // The fused finish (lvl0_finish_kernel): the global passes stop above the low FIN_LOW_BITS key bits.  depth0 = the
    // symbols that lie wholly inside the top part -- what the members of a bucket are known to share.
    const int total_bits = ht ? ht_sb + docs.bits : w * bt + spare + docs.bits;
    int depth0 = 0;
    for (int j = 0; j < w; j++)
        if (spare + j * bt >= FIN_LOW_BITS) depth0++;
    if (ht) depth0 = (ht_sb - FIN_LOW_BITS) / ht->max_len;          // the whole symbols any top part holds at least
    bool fused = allow_fused && ctx.knobs.fused_finish && n0 == 0 && total_bits >= 2 * FIN_LOW_BITS && depth0 >= 1;
    bool fin_small_halo = false;                        // buckets of at most ten suffixes expected: the kernel with the halo of 32
    if (fused && !ctx.dry) {
        if (ctx.knobs.force_fused) {
            // (test knob: buckets of any size -- the large ones go to the rounds)
        } else if (ctx.plan_fused >= 0) {
            fused = ctx.plan_fused != 0;                 // (speculative build: as the build before)
            fin_small_halo = ctx.plan_fused == 2 && !ht;
        } else if (ht) {
            // (variable-length keys come with a wide window: few suffixes per bucket of the top part)
        } else {
            // expected members of a bucket, were the text uniform
            double top_codes = pow((double)term_first, depth0);
            const int part = (spare + (w - depth0) * bt) - FIN_LOW_BITS;   // bits of the next symbol inside the top part
            if (part > 0) top_codes *= std::max(1.0, (double)term_first / (double)(1u << (bt - part)));
            const double expected = (double)(longest ? longest : n02) / top_codes;
            fused = expected <= FIN_MAX_EXPECTED;
            fin_small_halo = expected <= 10.0;
            // skewed text behind a narrow window: many buckets would be handed to the rounds whole (measured on the
            // Zipf stand-in: 8.9 against 8.2 ms); the sample tells -- more than 1 in 20 of its suffixes sharing the
            // symbols of the top part with three others of the sample is natural language, random text has none
            if (fused && ctx.sample_n && sizeof(K) == 4)
                fused = (u64)ctx.sample_dup4[std::min(depth0 + 1, 8)] * 20u <= ctx.sample_n;
        }
    }
    const int low_bits = fused ? FIN_LOW_BITS : 0;
    if (g_trace && n0 == 0 && !ctx.dry)
        fprintf(stderr, "[east_hip] level-0 keys: %d-bit, w = %d x %d bits + %d spare%s, %d key bits, global passes from bit %d, fused finish %d%s, "
                        "top part holds %d symbols, segmented by document %d\n", (int)sizeof(K) * 8, w, bt, spare, ht ? " (variable-length)" : "",
                total_bits, low_bits, (int)fused, fin_small_halo ? " (halo 32)" : "", depth0, (int)(docs.seg.n_docs != 0));
    if (ht) w = fused ? depth0 : ht_sb / ht->max_len;     // the depth the rounds start from: what every key (its top part) holds at least
    if (n0 == 0) ctx.did_fused = fused ? (fin_small_halo ? 2 : 1) : 0;
    if (ctx.stats && n0 == 0) ctx.stats->fused_finish = fused;
    SortBufs<K> sb;
    // (one spare element each: the idle half serves as scratch after the sort)
    // (8 spare entries: the placement pass reads whole 16-byte groups; >= 64 so that the idle half can hold its scratch)
    const size_t n_alloc = (size_t)(n02 > 56 ? n02 : 56) + 8;
    for (int k = 0; k < 2; k++) { sb.keys[k] = ar.alloc<K>(n_alloc); sb.vals[k] = ar.alloc<u32>(n_alloc); }
    if (docs.bits) {
        const u32 n_dt = (n02 >> DOC_TILE_SHIFT) + 1;
        u32 *tile_doc = ar.alloc<u32>(n_dt);
        LAUNCH(ctx, doc_tiles_kernel, ceil_div_u32(n_dt, BLOCK), docs.doc_off, docs.n_docs, n_dt, tile_doc);
        docs.tile_doc = tile_doc;
    }
    const int r = n0 ? radix_sort_pairs<K, WindowSrc<K>>(ctx, sb, n02, total_bits, 0,
                                                         WindowSrc<K>{s8, n0, w, bt, spare, term_first, docs})
                : ht ? radix_sort_pairs<K, HtWindowGen<K>>(ctx, sb, n02, total_bits, low_bits,
                                                           HtWindowGen<K>{s8, n02, ht_sb, ht->enc, docs},
                                                           docs.bits ? total_bits - docs.bits : total_bits + RS_DB, docs.seg)
                     : radix_sort_pairs<K, TextWindowGen<K>>(ctx, sb, n02, total_bits, low_bits,
                                                             TextWindowGen<K>{s8, n02, w, bt, spare, term_first, docs},
                                                             // (suffixes in text order: only the document number is sorted)
                                                             docs.bits ? total_bits - docs.bits : total_bits + RS_DB, docs.seg);
    KeyNeqWindowIn<K> starts = KeyNeqWindowIn<K>::make(sb.keys[r], ht ? 0 : w, bt, spare, term_first);
    if (ht) { starts.ht_sb = ht_sb; starts.ht_wmin = w; starts.ht_dec = ht->dec; }
    const u32 *sorted_vals = sb.vals[r];
    u64 *keep = (u64 *)sb.keys[r ^ 1];                   // n02 + 1 bits, in the keys idle since the sort
    u64 *gstart_bits = ar.alloc<u64>(((size_t)n02 >> 6) + 2);  // fused finish: the first rank of every group left to the rounds
    u32 *idx = sb.vals[r ^ 1];                          // n02 + 1 entries, likewise
    u32 *names_g = s12 ? ar.alloc<u32>(n02) : nullptr;  // sample mode: the naming predicate as refined so far
    u32 *fail = ar.alloc<u32>(1);
    const bool fail_is_zero = ctx.zeroed_word != nullptr;               // (zeroed with the build's flag words: one fill less)
    if (fail_is_zero) { fail = ctx.zeroed_word; ctx.zeroed_word = nullptr; }
    const u32 gp = ceil_div_u32((u64)n02 + 1, BLOCK * PLACE_IPT);      // workgroups of the placement pass
    u32 *block_keep = ar.alloc<u32>(gp);
    const u32 nb = ceil_div_u32(gp, SCAN_TILE);
    u32 *block_sums = ar.alloc<u32>(nb);
    u32 *bad = ar.alloc<u32>(((size_t)n02 >> 5) + 2);   // groups with a repeat too long to compare directly (rare)
    uint8_t *xdep0 = ht || ctx.dry ? ar.alloc<uint8_t>((size_t)n02 + 16) : nullptr;   // variable-length keys: extra depth per rank
    if (!ctx.dry && !fail_is_zero) HIP_CHECK(hipMemsetAsync(fail, 0, sizeof(u32), ctx.stream));

    // ---- the whole sorted input as the first domain --------------------------------------
    // (a small input is better off with the direct ordering of much larger groups than with a round of ~50 launches)
    KgMark km;                                          // k-gram bucket starts ride along with the first placement pass
    // (repetitive text -- the sample says, or the build before did -- is not ordered directly however small it is: its tie
    // groups agree for hundreds of symbols, and every member would compare itself with every other to the end; the
    // persistent rounds take it, persist_rounds.h)
    const bool repetitive_text = n0 == 0 && !ctx.dry && ctx.knobs.lds_rounds && ctx.knobs.persist && n02 >= 4u * PR_CTL_WORDS &&
                                 (ctx.rep_n ? ctx.rep_dup * 2u > ctx.rep_n : ctx.plan_persist == 1);
    const bool small_input = n02 <= REFINE_SMALL_INPUT && !repetitive_text;
    if (kg_mark) kg_mark->k = kg_mark->kg && n0 == 0 && !ctx.dry && !small_input && !ht ? kg_mark->k : 0;   // (variable-length keys: the score side builds its tables itself)
    if (kg_mark && kg_mark->k > 0) {
        km = *kg_mark;
        km.k = std::min(km.k, w);
        km.bins = 1;
        for (int i = 0; i < km.k; i++) km.bins *= km.A;
        km.pairs = km.kg3 && km.k >= 2;
        km.by_rank = docs.seg.n_docs != 0;
        HIP_CHECK(hipMemsetAsync(km.kg, 0xFF, (size_t)(km.bins + 1) * km.n_docs * (km.pairs ? 8 : 4), ctx.stream));
        if (km.pairs) HIP_CHECK(hipMemsetAsync(km.kg3, 0xFF, (size_t)(km.bins / km.A + 1) * km.n_docs * sizeof(u32), ctx.stream));
        *kg_mark = km;
    }
    u32 m = n02, m_next = n02, h_fail = 0;              // (sizing run: as if everything were tied)
    FinishArgs<K> fa;
    const SegRanks sr{docs.seg.n_docs ? docs.seg.doc_off : nullptr, docs.seg.n_docs};
    if (fused) {
        fa.keys = sb.keys[r]; fa.vals = sorted_vals; fa.m = n02; fa.s8 = s8;
        fa.w = w; fa.b = bt; fa.spare = spare; fa.low_bits = low_bits;
        fa.inv_b = (65536u + (u32)bt - 1u) / (u32)bt;
        fa.rep_t = starts.rep_t; fa.ones = starts.ones; fa.highs = starts.highs;
        fa.top_mask = 0;
        for (int j = 0; j < w; j++)
            if (spare + j * bt >= low_bits) fa.top_mask |= (K)(((K)1 << bt) - 1) << (spare + j * bt);
        fa.kg_top = spare + (w - km.k) * bt < low_bits;
        fa.order_g = sa12; fa.lcp_g = lcp_out; fa.keep = keep; fa.gstart = gstart_bits;
        fa.block_keep = block_keep; fa.fail = fail; fa.kg_bad = ctx.kg_bad ? ctx.kg_bad : fail;
        fa.km = small_input ? KgMark() : km;
        if (ht) { fa.ht_sb = ht_sb; fa.ht_wmin = w; fa.ht_dec = ht->dec; fa.xdep0 = xdep0; }
        if (sr.n_docs) { fa.km.by_rank = 1; fa.km.doc_off = sr.doc_off; fa.km.n_docs = sr.n_docs; }
        if (!ctx.dry) {
            HIP_CHECK(hipMemsetAsync(keep, 0, (((size_t)n02 >> 6) + 2) * sizeof(u64), ctx.stream));
            HIP_CHECK(hipMemsetAsync(gstart_bits, 0, (((size_t)n02 >> 6) + 2) * sizeof(u64), ctx.stream));
        }
    }
    auto place = [&](int mode) {
        if (fused) {
            if (mode != 0) throw FusedAbort();
            const bool sg = sr.n_docs != 0;
            if (ht && small_input && sg) LAUNCH_NAMED(ctx, "lvl0_finish_kernel", (lvl0_finish_kernel<K, true, 64, true, true>), gp, fa);
            else if (ht && small_input) LAUNCH_NAMED(ctx, "lvl0_finish_kernel", (lvl0_finish_kernel<K, true, 64, true>), gp, fa);
            else if (ht && sg) LAUNCH_NAMED(ctx, "lvl0_finish_kernel", (lvl0_finish_kernel<K, false, 64, true, true>), gp, fa);
            else if (ht) LAUNCH_NAMED(ctx, "lvl0_finish_kernel", (lvl0_finish_kernel<K, false, 64, true>), gp, fa);
            else if (small_input && sg) LAUNCH_NAMED(ctx, "lvl0_finish_kernel", (lvl0_finish_kernel<K, true, 64, false, true>), gp, fa);
            else if (small_input) LAUNCH_NAMED(ctx, "lvl0_finish_kernel", (lvl0_finish_kernel<K, true, 64>), gp, fa);
            else if (fin_small_halo && sg) LAUNCH_NAMED(ctx, "lvl0_finish_kernel", (lvl0_finish_kernel<K, false, 32, false, true>), gp, fa);
            else if (fin_small_halo) LAUNCH_NAMED(ctx, "lvl0_finish_kernel", (lvl0_finish_kernel<K, false, 32>), gp, fa);
            else if (sg) LAUNCH_NAMED(ctx, "lvl0_finish_kernel", (lvl0_finish_kernel<K, false, 64, false, true>), gp, fa);
            else LAUNCH_NAMED(ctx, "lvl0_finish_kernel", (lvl0_finish_kernel<K, false, 64>), gp, fa);
        } else if (ht && small_input)
            LAUNCH_NAMED(ctx, "lvl0_place_kernel", (lvl0_place_kernel<K, true, false, true>), gp, starts, sorted_vals, n02, s8, n0, w,
                         bt, spare, sa12, names_g, lcp_out, keep, block_keep, fail, LongRepeats{bad, mode}, KgMark(), xdep0, sr);
        else if (ht && mode == 0)
            LAUNCH_NAMED(ctx, "lvl0_place_kernel", (lvl0_place_kernel<K, false, true, true>), gp, starts, sorted_vals, n02, s8, n0, w,
                         bt, spare, sa12, names_g, lcp_out, keep, block_keep, fail, LongRepeats{bad, mode}, KgMark(), xdep0, sr);
        else if (ht)
            LAUNCH_NAMED(ctx, "lvl0_place_kernel", (lvl0_place_kernel<K, false, false, true>), gp, starts, sorted_vals, n02, s8, n0, w,
                         bt, spare, sa12, names_g, lcp_out, keep, block_keep, fail, LongRepeats{bad, mode}, KgMark(), xdep0, sr);
        else if (small_input)
            LAUNCH_NAMED(ctx, "lvl0_place_kernel", (lvl0_place_kernel<K, true, false>), gp, starts, sorted_vals, n02, s8, n0, w,
                         bt, spare, sa12, names_g, lcp_out, keep, block_keep, fail, LongRepeats{bad, mode}, KgMark(), (uint8_t *)nullptr, sr);
        else if (mode == 0)
            LAUNCH_NAMED(ctx, "lvl0_place_kernel", (lvl0_place_kernel<K, false, true>), gp, starts, sorted_vals, n02, s8, n0, w,
                         bt, spare, sa12, names_g, lcp_out, keep, block_keep, fail, LongRepeats{bad, mode}, km, (uint8_t *)nullptr, sr);
        else
            LAUNCH_NAMED(ctx, "lvl0_place_kernel", (lvl0_place_kernel<K, false, false>), gp, starts, sorted_vals, n02, s8, n0, w,
                         bt, spare, sa12, names_g, lcp_out, keep, block_keep, fail, LongRepeats{bad, mode}, KgMark(), (uint8_t *)nullptr, sr);
        if (mode == 1 || ctx.dry) return;
        if (ctx.spec_rounds && mode == 0 && n0 == 0) {
            // speculative build: the host goes on as if nothing were left in large groups; the counts are
            // put next to the build's other flags and read at the end (build_common repeats the build if
            // they are not zero)
            LAUNCH(ctx, spec_counts_kernel, ceil_div_u32(gp, SPEC_COUNT_TILE), (const u32 *)block_keep, gp, (const u32 *)fail,
                   ctx.spec_out);
            m_next = 0;
            h_fail = 0;
            return;
        }
        LAUNCH(ctx, (scan_reduce_kernel<ArrIn>), nb, ArrIn{block_keep}, gp, block_sums);
        std::vector<u32> h_sums(nb);
        HIP_CHECK(hipMemcpyAsync(h_sums.data(), block_sums, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx.stream));
        HIP_CHECK(hipMemcpyAsync(&h_fail, fail, 4, hipMemcpyDeviceToHost, ctx.stream));
        HIP_CHECK(sync_stream(ctx.stream));
        m_next = 0;
        for (u32 x : h_sums) m_next += x;
        if (ctx.stats && n0 == 0 && mode == 0) { ctx.stats->first_kept = m_next; ctx.stats->first_n = n02; }
        if (g_trace)
            fprintf(stderr, "[east_hip] level-0 (%s, %u elements, w = %d): %u in large groups%s\n", n0 ? "sample" : "all suffixes",
                    n02, w, m_next, h_fail ? ", a repeat too long to order directly" : "");
    };
    place(0);
    if (ctx.dry) LAUNCH(ctx, (scan_reduce_kernel<ArrIn>), nb, ArrIn{block_keep}, gp, block_sums);
    bool long_repeats = false;                          // small groups were handed to the rounds
    if (h_fail) {
        if (fused) throw FusedAbort();                   // (duplicated passages: the mark + commit passes want the sorted pairs)
        long_repeats = true;
        if (ctx.stats) ctx.stats->long_repeats++;
        // duplicated passages inside small groups: put the domain back, mark those groups, place the others
        LAUNCH(ctx, (dc3_refine_restore_kernel<KeyNeqWindowIn<K>>), g02, sorted_vals, starts, (const u32 *)nullptr, n02, sa12,
               names_g);
        HIP_CHECK(hipMemsetAsync(fail, 0, sizeof(u32), ctx.stream));
        HIP_CHECK(hipMemsetAsync(bad, 0, (((size_t)n02 >> 5) + 2) * sizeof(u32), ctx.stream));
        place(1);
        place(2);
        if (h_fail) east_throw(EAST_HIP_ERR_INTERNAL, "level 0: undecided comparison after the marking pass");
    }
    if (!ctx.dry && m_next == 0) {                      // everything is in place (and the LCP table written)
        if (ctx.stats) ctx.stats->levels_resolved++;
        return true;
    }

    // ---- refinement rounds on the members of large groups ---------------------------------
    // (ctx.lean: the device is short of memory for the rounds' buffers -- straight on to the recursion / DC3)
    // the placement pass's verdict is kept: the LCP entries of everything it placed are final
    // (the ranks whose entries the rounds do NOT write -- those that go through prefix-doubling rounds, whose keys are
    // names -- are marked when the rounds switch over, and only they are computed at the end)
    // (the ranks whose entries have to be computed at the end: the slots of the domain of the round that switched over)
    u32 *lcp_redo = lcp_out ? ar.alloc<u32>((size_t)n02 + 1) : nullptr;
    u32 redo_n = 0;
    bool redo_pending = false, redo_hinted = false;
    bool done = false;
    bool lcp_from_rounds = true;                        // the rounds write the LCP entries of what they place (symbol windows)
    if (!ctx.lean) {
        const size_t mark_rounds = ar.mark();
        const u32 cap = n02 + 1;                        // natural-language text: most suffixes can be in large groups
        u32 *ebuf[3], *sbuf[2], *fbuf[2];
        for (auto &e : ebuf) e = ar.alloc<u32>(cap);
        for (auto &e : sbuf) e = ar.alloc<u32>(cap);
        for (auto &e : fbuf) e = ar.alloc<u32>(cap);
        u32 *gstart = ar.alloc<u32>(cap), *group = ar.alloc<u32>(cap);
        SortBufs<u64> rb;
        for (int k = 0; k < 2; k++) { rb.keys[k] = ar.alloc<u64>(cap); rb.vals[k] = ar.alloc<u32>(cap); }
        u32 *name_of = n0 == 0 ? ar.alloc<u32>(n02) : nullptr;      // all-suffix mode: names for prefix doubling
        // variable-length first-level keys: what the group of a domain position shares beyond `depth` (see HtKeys)
        uint8_t *xbuf[2] = {nullptr, nullptr};
        if (ht || ctx.dry) for (auto &x : xbuf) x = ar.alloc<uint8_t>((size_t)cap + 16);
        int x_dom = 0;
        bool have_x = false;                            // xbuf[x_dom] describes the current domain
        // the in-LDS round (lds_group_sort.h): what each workgroup took, and the compaction of the rest
        uint2 *cover = ar.alloc<uint2>((size_t)cap / LG_CHUNK + 2);
        u32 *sub_idx = ar.alloc<u32>((size_t)cap + 1), *full_idx = ar.alloc<u32>(cap);
        u32 *rest_cnt = ar.alloc<u32>((size_t)cap / LG_CHUNK + 3), *rest_pre = ar.alloc<u32>((size_t)cap / LG_CHUNK + 3);
        if (ctx.dry) {                                  // sizing run: the transient buffers of one round (on top of all of the above)
            device_scan<BitIn, false>(ctx, BitIn{keep}, n02 + 1, idx);
            (void)radix_sort_pairs<u64>(ctx, rb, cap, 8);
        }
        bool doubling = false;
        PersistCap persist_caps = ctx.knobs.lds_rounds && ctx.knobs.persist && !ctx.dry ? persist_capacity() : PersistCap();
        if (ctx.knobs.persist_max_wgs > 0) {             // (tests / experiments: a smaller grid -- several tiles per workgroup on small inputs)
            const u32 cap_wgs = (u32)ctx.knobs.persist_max_wgs;
            persist_caps.one_per_cu = std::min(persist_caps.one_per_cu, cap_wgs);
            persist_caps.large = std::min(persist_caps.large, cap_wgs);
        }
        u32 persist_cap = persist_caps.one_per_cu;
        u32 depth = fused ? (u32)depth0 : (u32)w;        // what the members of a group are known to share
        const u32 *elem = fused ? (const u32 *)sa12 : sorted_vals, *slot = nullptr, *flag = nullptr;
        int e_dom = -1, s_dom = 0, f_dom = 0;           // which of the rotating buffers hold the domain
        bool have_idx = false;                          // idx = exclusive scan of keep over the domain
        int stalled = 0;                                // rounds in a row that placed (almost) nothing
        for (int round = 0; !ctx.dry; round++) {
            if (m_next == 0) { done = true; break; }
            // Strings of a few words dissolve within their own length, a round takes 6-12 symbols off.  A domain
            // that stops shrinking is a long repeat: all-suffix mode goes over to prefix doubling (below), the
            // sample mode of DC3 stops here (every round would cost the same again) and recurses on the names.
            stalled = (round > 0 && m_next > m - m / 32) ? stalled + 1 : 0;
            if (round == REFINE_MAX_ROUNDS || (stalled == 2 && !doubling && !name_of)) break;
            const u32 gm = ceil_div_u32((u64)m + 1, BLOCK);
            // A domain that fits the chip at one tile per resident workgroup can be finished by ONE launch (persist_rounds.h):
            // prefix doubling with the tiles kept in LDS, a grid barrier per round.  It is taken where the rounds below would
            // go over to prefix doubling -- a domain that shrinks slowly: long repeats -- and for the FIRST domain of text
            // the planning sample calls repetitive (most of 8 192 consecutive suffixes share 8 symbols with three others of
            // the sample: the reference's worst case, 100 identical strings; a speculative build: as the build before).
            // Natural language is better off launch by launch: its domains shrink eightfold per round and the endgame orders
            // what is left directly (measured on 0.25 .. 4 MiB of prose: 0.29-0.89 ms against 0.31-0.98 through doubling).
            // The launch gives up -- having changed nothing the rounds below rely on -- when a tie group is longer than a tile
            // takes; a later, smaller domain may fit.
            // (the launch needs the name of every placed suffix first -- a scatter over all n02 ranks, 2.1 ms for the 94 M
            // symbols of the Zipf stand-in: only where the domain is a good part of the input, or the input small)
            const bool slow = round > 0 && m_next > m / 2;          // slow shrinking = long repeats
            const bool repetitive = repetitive_text;
            // (a domain of more tiles than the device holds workgroups: the same rounds with the tiles' state in global
            // memory, several tiles per workgroup -- refine_persist2_kernel)
            const bool try_persist = persist_cap > 0 && name_of && !doubling && n02 >= 4u * PR_CTL_WORDS &&
                                     (m_next <= persist_cap * LG_CHUNK || (persist_caps.large > 0 && ctx.knobs.persist_large)) &&
                                     (n02 <= PERSIST_SMALL_INPUT || n02 / 8u <= m_next) &&
                                     // (... or whose first window -- wide enough to tell random suffixes apart -- left nineteen of
                                     // twenty suffixes tied: copies further apart than the sample is long)
                                     (round == 0 ? repetitive || m_next >= n02 - n02 / 20u : slow);
            if (name_of && !doubling && slow && !try_persist) {
                // slow shrinking = long repeats: from here on the depth doubles every round (see above)
                doubling = true;
                lcp_from_rounds = false;                // (names instead of symbols: the seams' entries are computed at the end)
                redo_pending = lcp_redo != nullptr;     // (the compaction below lists the slots of the members still open)
                LAUNCH(ctx, dc3_names_init_kernel, ceil_div_u32(n02, BLOCK), (const u32 *)sa12, n02, name_of);
                device_scan<ArrIn, true>(ctx, ArrIn{flag}, m, group);
                LAUNCH(ctx, dc3_group_starts_kernel, gm, flag, (const u32 *)group, slot, m, gstart);
                LAUNCH(ctx, dc3_names_update_kernel, gm, elem, (const u32 *)group, (const u32 *)gstart, m, (const u64 *)keep,
                       name_of);
                if (g_trace) fprintf(stderr, "[east_hip]   switching to prefix doubling at depth %u\n", depth);
            }
            if (!have_idx) device_scan<PopIn, false>(ctx, PopIn{keep, (m >> 6) + 1u}, (m >> 6) + 2u, idx);
            // compact the members of large groups, number their groups, sort by (group, next window)
            const int e_c = (e_dom + 4) % 3, e_out = (e_dom + 5) % 3;      // the two buffers the domain is not in
            u32 *slot_c = sbuf[s_dom ^ 1];
            const u32 gc = ceil_div_u32((u64)m + 1, BLOCK * COMPACT_IPT);
            // (xdep: computed from the keys by the first compaction, copied along by the later ones; prefix doubling rounds
            // go by the common depth alone)
            const uint8_t *x_in = have_x ? (const uint8_t *)xbuf[x_dom] : !slot && ht ? (const uint8_t *)xdep0 : (const uint8_t *)nullptr;
            uint8_t *x_out = ht && !doubling ? xbuf[x_dom ^ 1] : (uint8_t *)nullptr;
            if (!slot && fused)
                LAUNCH_NAMED(ctx, "dc3_refine_compact_kernel", (dc3_refine_compact_kernel<BitStarts>), gc, elem,
                             BitStarts{gstart_bits}, slot, BitIn{keep}, BitRank{keep, idx}, m, slot_c, ebuf[e_c], gstart,
                             (u32 *)nullptr, 0, 0, 0, x_in, x_out);
            else if (!slot)
                LAUNCH_NAMED(ctx, "dc3_refine_compact_kernel", (dc3_refine_compact_kernel<KeyNeqWindowIn<K>>), gc, elem,
                             starts, slot, BitIn{keep}, BitRank{keep, idx}, m, slot_c, ebuf[e_c], gstart, lcp_out, w, bt, spare,
                             x_in, x_out);
            else
                LAUNCH_NAMED(ctx, "dc3_refine_compact_kernel", (dc3_refine_compact_kernel<FlagArrIn>), gc, elem,
                             FlagArrIn{flag}, slot, BitIn{keep}, BitRank{keep, idx}, m, slot_c, ebuf[e_c], gstart,
                             (u32 *)nullptr, 0, 0, 0, x_in, x_out);
            if (x_out) { x_dom ^= 1; have_x = true; } else have_x = false;
            if (redo_pending) {
                HIP_CHECK(hipMemcpyAsync(lcp_redo, slot_c, (size_t)m_next * 4, hipMemcpyDeviceToDevice, ctx.stream));
                redo_n = m_next;
                redo_pending = false;
            }
            const uint8_t *xdep = have_x ? (const uint8_t *)xbuf[x_dom] : (const uint8_t *)nullptr;
            m = m_next;
            if (try_persist) {
                // (the second copy of the names and the control words: the round sort's buffers, idle until a sort runs)
                u32 *name1 = (u32 *)rb.keys[0], *ctl = rb.vals[0];
                HIP_CHECK(hipMemsetAsync(ctl, 0, PR_CTL_WORDS * sizeof(u32), ctx.stream));
                LAUNCH(ctx, persist_names_init_kernel, ceil_div_u32(n02, BLOCK), (const u32 *)sa12,
                       !slot && !fused ? sorted_vals : (const u32 *)nullptr, (const u64 *)keep, n02, name_of, name1);
                const int name_bits = bit_width_u32(n02 > 1 ? n02 - 1 : 1);
                const PrArgs pa{ebuf[e_c], gstart, slot_c, m, depth, name_bits, name_of, name1, sa12, lcp_redo ? lcp_out : (u32 *)nullptr, ctl};
                const u32 tiles = ceil_div_u32(m, LG_CHUNK);
                // (the large form: the second copy of the group bounds, the tiles' ranges and "changed" flags in buffers of the
                // launch-by-launch rounds that are idle here)
                const Pr2Args pa2{ebuf[e_c], slot_c, gstart, full_idx, m, depth, name_bits, name_of, sa12, lcp_redo ? lcp_out : (u32 *)nullptr,
                                  ctl, cover, rest_cnt, tiles};
                u32 h_ctl[4] = {0, 0, 0, 0};
                {
                    std::lock_guard<std::mutex> lock(g_persist_mutex);
                    const bool force_large = ctx.knobs.persist_force_large && persist_caps.large > 0;      // (tests: the large form on small domains)
                    if (tiles <= persist_caps.one_per_cu && !force_large) LAUNCH_BLOCK(ctx, refine_persist_kernel<4>, tiles, LG_THREADS, pa);
                    // (between one and two tiles per CU the resident form at two workgroups per CU -- 64 registers, spills, and
                    // with the barrier's fences 1.07 ms for the 488 tiles of the worst case at n = 10^4 -- loses to the large
                    // form at one workgroup per CU, 0.8 ms: it is not used)
                    else LAUNCH_BLOCK(ctx, refine_persist2_kernel, std::min(tiles, persist_caps.large), LG_THREADS, pa2);
                    HIP_CHECK(hipMemcpyAsync(h_ctl, ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, ctx.stream));
                    HIP_CHECK(sync_stream(ctx.stream));
                }
                if (h_ctl[PR_CTL_ABORT])
                    east_throw(EAST_HIP_ERR_INTERNAL, h_ctl[PR_CTL_ABORT] == 2 ? "persistent rounds: groups still open after the last round"
                                                                                : "persistent rounds: a grid barrier timed out");
                if (g_trace)
                    fprintf(stderr, "[east_hip]   round %d: persistent launch over a domain of %u at depth %u: %s\n", round, m, depth,
                            h_ctl[PR_CTL_BAIL] ? "a group too long for a tile, the rounds go on launch by launch" : "done");
                if (h_ctl[PR_CTL_BAIL] && slow) persist_cap = 0;     // (long groups that do not split: prefix doubling launch by launch)
#ifdef PR_STAMPS
                if (g_trace && !h_ctl[PR_CTL_BAIL]) {
                    std::vector<unsigned long long> st(8 * PR_MAX_ROUNDS);
                    HIP_CHECK(hipMemcpy(st.data(), ctl + PR_CTL_STAMPS, st.size() * 8, hipMemcpyDeviceToHost));
                    for (u32 r = 0; r < h_ctl[PR_CTL_ROUNDS]; r++) {
                        auto d = [&](int x, int y) { return (double)(long long)(st[r * 8 + x] - st[r * 8 + y]) * 0.01; };
                        fprintf(stderr, "[east_hip]     round %u (block 0): gather+test %.2f us, sort %.2f, bounds %.2f, held loads %.2f, names+stores %.2f, count %.2f, barrier %.2f\n",
                                r, d(1, 0), d(2, 1), d(5, 2), d(6, 5), d(7, 6), d(3, 7), d(4, 3));
                    }
                    const u32 nwg = ceil_div_u32(m, LG_CHUNK);
                    std::vector<unsigned long long> ab(2 * (size_t)nwg);
                    HIP_CHECK(hipMemcpy(ab.data(), ctl + PR_CTL_WORDS, ab.size() * 8, hipMemcpyDeviceToHost));
                    if (h_ctl[PR_CTL_ROUNDS] > 3) {
                        unsigned long long t0 = ~0ull;
                        for (u32 b = 0; b < nwg; b++) t0 = std::min(t0, ab[2 * b]);
                        fprintf(stderr, "[east_hip]     round 3, arrival / departure of every 16th block (us after the first arrival):");
                        for (u32 b = 0; b < nwg; b += 16) fprintf(stderr, " %u:%.1f/%.1f", b, (ab[2 * b] - t0) * 0.01, (ab[2 * b + 1] - t0) * 0.01);
                        double mx = 0; u32 who = 0;
                        for (u32 b = 0; b < nwg; b++) if ((ab[2 * b] - t0) * 0.01 > mx) { mx = (ab[2 * b] - t0) * 0.01; who = b; }
                        fprintf(stderr, "\n[east_hip]     last arrival: block %u at %.1f us\n", who, mx);
                    }
                }
#endif
                if (!h_ctl[PR_CTL_BAIL]) {
                    if (ctx.stats) {
                        ctx.stats->refine_rounds += h_ctl[PR_CTL_ROUNDS];
                        ctx.stats->persist_rounds += h_ctl[PR_CTL_ROUNDS];
                        ctx.stats->lds_sorted += (i64)m * h_ctl[PR_CTL_ROUNDS];
                    }
                    // the LCP entries of the domain's ranks: compared on the text once the suffix array stands (below)
                    if (lcp_redo) {
                        HIP_CHECK(hipMemcpyAsync(lcp_redo, slot_c, (size_t)m * 4, hipMemcpyDeviceToDevice, ctx.stream));
                        redo_n = m;
                        lcp_from_rounds = false;
                        redo_hinted = true;
                    }
                    if (round == 0 && n0 == 0) ctx.did_persist = 1;
                    m_next = 0;
                    done = true;
                    break;
                }
            }
            bool have_group = false;                    // group[] = inclusive scan of gstart: the groups' numbers
            auto number_groups = [&]() {
                if (!have_group) device_scan<ArrIn, true>(ctx, ArrIn{gstart}, m, group);
                have_group = true;
            };
            // (every group here has more than REFINE_SMALL_GROUP members -- at least 2 once groups with long repeats
            // were handed over: a bound on their number saves a read-back)
            const int gbits = bit_width_u32(m / (long_repeats ? 2u : REFINE_SMALL_GROUP + 1u) + 1);
            const u32 gt = ceil_div_u32((u64)m + 1, BLOCK);
            // One round's sort: the members of every group ordered by their round key -- the next window of symbols, or
            // (prefix doubling) the name of the suffix `depth` symbols further on.  Groups that fit a workgroup's LDS are
            // sorted there, keys, sort and write-back in one launch (lds_group_sort.h); the rest is compacted and takes
            // the global radix sort by (group, key).
            // (the classification of the next domain rides along with the in-LDS round, see LgClassify: symbol windows, the
            // usual limits, no groups with long repeats known)
            // (once groups with long repeats are known to exist the direct ordering of large groups only burns time: every
            // member of such a group compares REFINE_ENDGAME_LEN symbols with every other before it gives up)
            const bool endgame = m <= REFINE_ENDGAME_DOMAIN && !long_repeats && !repetitive_text;
            const bool fuse_cls = ctx.knobs.lds_rounds && ctx.knobs.fused_classify && !doubling && !endgame && !long_repeats;
            u32 round_left = m;                         // what the in-LDS round left to the global sort
            auto sort_round = [&](bool names, int w2, const KeyNeqWindowIn<u64> &f) {
                const int kbits = names ? w2 : w2 * bt;     // (names: w2 = the bits of a name)
                u32 *lcp_r = names ? (u32 *)nullptr : lcp_out;
                u32 m_left = m;
                if (ctx.knobs.lds_rounds) {
                    if (fuse_cls) HIP_CHECK(hipMemsetAsync(keep, 0, (((size_t)m >> 6) + 2) * sizeof(u64), ctx.stream));
                    LAUNCH_BLOCK(ctx, refine_lds_sort_kernel, ceil_div_u32(m, LG_CHUNK), LG_THREADS, s8, (const u32 *)ebuf[e_c],
                                 (const u32 *)gstart, (const u32 *)slot_c, m, n0, depth, w2, bt, term_first, (u64)f.rep_t,
                                 (u64)f.ones, (u64)f.highs, sa12, names_g, ebuf[e_out], fbuf[f_dom ^ 1], lcp_r, cover,
                                 names ? (const u32 *)name_of : (const u32 *)nullptr,
                                 fuse_cls ? LgClassify{keep, fail, (u32)REFINE_SMALL_GROUP, (u32)RESOLVE_MAX_LEN, names ? nullptr : xdep}
                                          : LgClassify{nullptr, nullptr, 0u, 0u, names ? nullptr : xdep});
                    // what is left: counted per chunk from cover[] (no pass over the elements)
                    const u32 n_chunks = ceil_div_u32(m, LG_CHUNK);
                    LAUNCH(ctx, lg_rest_count_kernel, ceil_div_u32(n_chunks + 1, BLOCK), (const uint2 *)cover, m, n_chunks, rest_cnt);
                    device_scan<ArrIn, false>(ctx, ArrIn{rest_cnt}, n_chunks + 1, rest_pre);
                    HIP_CHECK(hipMemcpyAsync(&m_left, rest_pre + n_chunks, 4, hipMemcpyDeviceToHost, ctx.stream));
                    HIP_CHECK(sync_stream(ctx.stream));
                    if (ctx.stats) ctx.stats->lds_sorted += m - m_left;
                    if (g_trace) fprintf(stderr, "[east_hip]   round %d: %u of %u sorted in LDS\n", round, m - m_left, m);
                }
                round_left = m_left;
                if (m_left == 0) return;
                const u32 *elems_s = ebuf[e_c];             // what goes through the global sort: the domain, or its rest
                const u32 *full = nullptr;
                int gb = gbits;
                if (m_left == m) {
                    number_groups();
                    // (the in-LDS round ran and took nothing: every group is longer than a tile takes -- the same bound on
                    // their number as below: a run of one letter is ONE group of 16 M, whose number was sorted on 24 bits)
                    if (ctx.knobs.lds_rounds) gb = std::min(gbits, bit_width_u32(m_left / (LG_MAX_GROUP + 1u) + 1u));
                } else {
                    // the rest, compacted (elements, group starts, where they came from), its groups numbered -- every
                    // one of them has more than LG_MAX_GROUP members, which bounds their number
                    u32 *sub_elem = sub_idx, *sub_gstart = rb.vals[1];           // (rb.vals[1]: idle until the sort's first pass)
                    LAUNCH(ctx, lg_rest_compact_kernel, gt, LgUncovered{cover, m}, (const u32 *)rest_pre, (const u32 *)ebuf[e_c],
                           (const u32 *)gstart, m, sub_elem, sub_gstart, full_idx);
                    device_scan<ArrIn, true>(ctx, ArrIn{sub_gstart}, m_left, group);
                    elems_s = sub_elem;
                    full = full_idx;
                    gb = std::min(gbits, bit_width_u32(m_left / (LG_MAX_GROUP + 1u) + 1u));
                }
                const u32 gl = ceil_div_u32(m_left, BLOCK);
                if (names)
                    LAUNCH(ctx, dc3_double_keys_kernel, gl, (const u32 *)name_of, elems_s, (const u32 *)group, m_left, depth, w2,
                           rb.keys[0], rb.vals[0]);
                else
                    LAUNCH(ctx, dc3_refine_keys_kernel, gl, s8, elems_s, (const u32 *)group, m_left, n0, depth, w2, bt, term_first,
                           rb.keys[0], rb.vals[0], xdep, full);
                const int rr = radix_sort_pairs<u64>(ctx, rb, m_left, gb + kbits);
                LAUNCH(ctx, dc3_refine_writeback_kernel, gl, (const u64 *)rb.keys[rr], (const u32 *)rb.vals[rr], (const u32 *)slot_c,
                       (const u32 *)ebuf[e_c], m_left, f.rep_t, f.ones, f.highs, sa12, names_g, ebuf[e_out], fbuf[f_dom ^ 1], lcp_r,
                       depth, names ? 0 : w2, bt, full, names ? (const uint8_t *)nullptr : xdep);
            };
            if (doubling) {
                sort_round(true, bit_width_u32(n02 > 1 ? n02 - 1 : 1), KeyNeqWindowIn<u64>{nullptr, 0, 0, 0});
                // the members' new names: where their (possibly split) group now starts
                device_scan<ArrIn, true>(ctx, ArrIn{fbuf[f_dom ^ 1]}, m, group);
                LAUNCH(ctx, dc3_group_starts_kernel, gt, (const u32 *)fbuf[f_dom ^ 1], (const u32 *)group, (const u32 *)slot_c, m,
                       gstart);
                LAUNCH(ctx, dc3_names_update_kernel, gt, (const u32 *)ebuf[e_out], (const u32 *)group, (const u32 *)gstart, m,
                       (const u64 *)nullptr, name_of);
                depth *= 2;
            } else {
                // (13 bits: room for the group numbers inside a workgroup's tile of the in-LDS round)
                int w2 = std::min(12, (64 - std::max(gbits, 13)) / bt);
                if (getenv("EAST_HIP_ROUND_W2")) w2 = std::max(1, std::min(w2, atoi(getenv("EAST_HIP_ROUND_W2"))));   // (experiments)
                if (w2 < 1) break;
                sort_round(false, w2, KeyNeqWindowIn<u64>::make(nullptr, w2, bt, 0, term_first));
                depth += (u32)w2;
            }
            e_dom = e_out; s_dom ^= 1; f_dom ^= 1;
            elem = ebuf[e_dom]; slot = sbuf[s_dom]; flag = fbuf[f_dom];
            if (ctx.stats) ctx.stats->refine_rounds++;
            // the new, smaller domain: place what is untied or in small groups now, count the rest
            // a small domain is finished by direct ordering of much larger groups (a round costs ~60 launches
            // however few suffixes are left); the comparisons are kept short so that the work stays bounded
            auto classify = [&](int mode, bool rest_only = false) {
                if (!rest_only || round_left > 0)       // (rest_only: the in-LDS round has classified its own tiles)
                    LAUNCH(ctx, dc3_refine_classify_kernel, gt, elem, FlagArrIn{flag}, slot, m, s8, n0, depth, sa12, names_g, keep,
                           // (doubling rounds with long repeats known, once the depth is beyond what a direct comparison may
                           // look at: the members of a small group would compare REFINE_DOUBLING_LEN symbols, give up, and do it
                           // again next round -- 19 of the 53 ms of a 16 M-symbol Fibonacci string; the next round's name look-up
                           // decides them: only what is alone in its group is placed)
                           fail, doubling && mode != 0 && depth >= REFINE_DOUBLING_LEN ? 1u : endgame ? (u32)REFINE_ENDGAME_GROUP : (u32)REFINE_SMALL_GROUP,
                           // (doubling rounds with long repeats known -- mark + commit, an undecided group simply stays for the
                           // next round, which doubles the depth by one name look-up per member: comparing 2 048 symbols of text
                           // for it first was 60 of the 96 ms of a 16 M-symbol Fibonacci string)
                           // (... and the first attempt of a doubling round is short as well: in a Fibonacci string the members of
                           // small groups DO differ within 2 048 symbols, and found that out 256 dependent loads at a time, every
                           // round -- 0.73 ms a launch)
                           endgame ? (u32)REFINE_ENDGAME_LEN : doubling ? (u32)REFINE_DOUBLING_LEN : (u32)RESOLVE_MAX_LEN,
                           doubling ? name_of : (u32 *)nullptr,
                           LongRepeats{bad, mode}, lcp_out, rest_only ? (const uint2 *)cover : (const uint2 *)nullptr,
                           doubling ? (const uint8_t *)nullptr : xdep);
                if (mode == 1) return;
                device_scan<PopIn, false>(ctx, PopIn{keep, (m >> 6) + 1u}, (m >> 6) + 2u, idx);
                have_idx = true;
                HIP_CHECK(hipMemcpyAsync(&m_next, idx + ((m >> 6) + 1u), 4, hipMemcpyDeviceToHost, ctx.stream));
                HIP_CHECK(hipMemcpyAsync(&h_fail, fail, 4, hipMemcpyDeviceToHost, ctx.stream));
                HIP_CHECK(sync_stream(ctx.stream));
                if (g_trace)
                    fprintf(stderr, "[east_hip]   round %d: domain %u, %u still in large groups, depth %u%s\n", round, m, m_next,
                            depth, h_fail ? ", a repeat too long to order directly" : "");
            };
            if (!long_repeats) classify(0, fuse_cls);   // (once such groups are known to exist: mark + place right away)
            if (h_fail || long_repeats) {               // as in the first domain: restore, mark, place the rest
                if (h_fail) LAUNCH(ctx, (dc3_refine_restore_kernel<FlagArrIn>), gt, elem, FlagArrIn{flag}, slot, m, sa12, names_g);
                HIP_CHECK(hipMemsetAsync(fail, 0, sizeof(u32), ctx.stream));
                HIP_CHECK(hipMemsetAsync(bad, 0, (((size_t)m >> 5) + 2) * sizeof(u32), ctx.stream));
                if (doubling && h_fail) {                         // (the abandoned pass named some suffixes by a place they do not get)
                    device_scan<ArrIn, true>(ctx, ArrIn{flag}, m, group);
                    LAUNCH(ctx, dc3_group_starts_kernel, gt, flag, (const u32 *)group, slot, m, gstart);
                    LAUNCH(ctx, dc3_names_update_kernel, gt, elem, (const u32 *)group, (const u32 *)gstart, m,
                           (const u64 *)nullptr, name_of);
                }
                long_repeats = true;
                classify(1);
                classify(2);
                if (h_fail) east_throw(EAST_HIP_ERR_INTERNAL, "level 0: undecided comparison after the marking pass");
            }
        }
        ar.release(mark_rounds);
    }
    if (done) {
        if (ctx.stats) ctx.stats->levels_resolved++;
        if (lcp_out && !lcp_from_rounds && redo_n)      // (prefix doubling: the entries of everything those rounds placed)
            LAUNCH(ctx, lvl0_lcp_text_list_kernel, ceil_div_u32(redo_n, BLOCK), s8, (const u32 *)sa12, (const u32 *)lcp_redo, redo_n,
                   lcp_out, lcp_capped, ctx.lcp_budget, redo_hinted);
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
        HIP_CHECK(sync_stream(ctx.stream));
        if (n_names == 0 || n_names > n02) east_throw(EAST_HIP_ERR_INTERNAL, "dc3: impossible name count");
    }
    LAUNCH(ctx, dc3_scatter_names_kernel, ceil_div_u32((u64)n02 + 3, BLOCK), (const u32 *)sa12, (const u32 *)names, n02,
           s12);
    return false;
}

// Width of the level-0 name window.  Widest window: names almost unique (sigma^w >= 64 n) within
// 64-bit keys; but if the window that still fits 32-bit keys leaves only a few per cent of ties
// (sigma^w >= 4 n), the cheaper sort wins and the tie resolution absorbs the difference.
// (doc_bits key bits are taken by the document number; n is then the length of the longest document)
static int lvl0_window(u32 n, int bt, u32 term_first, int doc_bits = 0)
{
    int w = 3;
    const int w_max = (64 - doc_bits) / bt < 12 ? (64 - doc_bits) / bt : 12;
    double reach = (double)term_first * term_first * term_first;
    while (w < w_max && reach < 64.0 * (double)n) { reach *= term_first; w++; }
    const int w32 = (32 - doc_bits) / bt;
    if (w32 >= 3 && w32 < w) {
        // the spare bits of the last digit hold the top bits of one more symbol: that many more buckets
        const int spare32 = lvl0_spare_bits(w32 * bt + doc_bits, 32, bt, w32);
        const double buckets = spare32 > 0 ? (double)((term_first >> (bt - spare32)) + 1u) : 1.0;
        if (pow((double)term_first, w32) * buckets >= 4.0 * (double)n) w = w32;
    }
    return w;
}

// The fast path for text: ALL n suffixes keyed by their first w symbols, one stable sort, the tied
// ones ordered directly / refined by further windows / by prefix doubling -- no sample, no ranks, no
// merge.  Returns false only in the rare cases it gives up (REFINE_MAX_ROUNDS, or no room for a
// window next to the document number); the caller then runs DC3, whose work is bounded whatever the input.
// Several documents (docs.bits > 0, longest = symbols of the longest one): the keys carry the document
// number on top, sa_out / lcp_out receive every document's tables side by side (lcp_out: the first
// entry of each document still has to be reset, lcp_doc_starts_kernel).
static bool window_suffix_sort(Ctx &ctx, const uint8_t *s8, u32 n, u32 term_first, u32 *sa_out, u32 *lcp_out,
                               u32 *lcp_capped, DocKey docs = DocKey(), u32 longest = 0, KgMark *kg_mark = nullptr)
{
    Arena &ar = *ctx.arena;
    const size_t mark = ar.mark();
    const int bt = bit_width_u32(term_first);
    const int kg_k_in = kg_mark ? kg_mark->k : 0;
    // Segmented sort (radix_sort.h: RsSeg): a handful to a few thousand LARGE documents keep their ranges in every pass and
    // the key holds text only -- 256 documents of 1 MiB: 6 symbols + 2 bits of the 7th in a 32-bit key instead of 4 + 4
    // bits; the buckets of the fused finish shrink from 28 suffixes to one, next to nothing stays tied.  Worth it while the
    // documents' own tiles and groups (the last group of a document is partly empty) stay within twice the flat count:
    // decided from n and the number of documents alone, so that the sizing run prices the same plan.
    const bool multi = docs.bits > 0;
    {
        const u32 flat_groups = ceil_div_u32(ceil_div_u32(n, RS_TILE), RS_GROUP);
        const bool can = multi && docs.n_docs <= RS_SEG_MAX_DOCS && (ctx.dry || docs.h_doc_off);
        // (documents of a histogram group -- 32 768 suffixes -- or more on average: measured on 256 MiB of word text, build
        // with / without: 256 x 1 MiB 7.3 / 7.8 ms, 1 024 x 256 KiB 6.9 / 7.3, 4 096 x 64 KiB 7.3 / 7.8, 8 192 x 32 KiB 7.9 / 8.4,
        // 16 384 x 16 KiB 7.9 / 7.7 -- short tiles and mostly empty groups)
        const u32 per_groups = getenv("EAST_HIP_SEG_DIV") ? (u32)std::max(1, atoi(getenv("EAST_HIP_SEG_DIV"))) : 1u;   // (experiments)
        // (... and five or more of them: the document number of two to four documents costs the key a bit or two, less than
        // the segments' short tiles cost the passes -- 2 x 32 MiB: 1.71 ms with, 1.68 without)
        if (can && ctx.knobs.seg_mode != 0 && (ctx.knobs.seg_mode == 1 || (docs.n_docs >= 5 && docs.n_docs <= flat_groups / per_groups + 1))) {
            RsSeg &seg = docs.seg;
            seg.n_docs = docs.n_docs;
            seg.doc_off = docs.doc_off;
            seg.shards = docs.n_docs < RS_TOTAL_SHARDS ? RS_TOTAL_SHARDS : 1;
            u32 *d_group0 = ar.alloc<u32>((size_t)docs.n_docs + 1);
            seg.doc_group0 = d_group0;
            if (ctx.dry) {
                seg.n_groups = flat_groups + docs.n_docs;
                seg.group_doc = ar.alloc<u32>(seg.n_groups);
                seg.big_docs = ar.alloc<u32>(docs.n_docs);
            } else {
                std::vector<u32> &hs = ctx.seg_host;        // (lives as long as the build: the copies below are asynchronous)
                hs.assign((size_t)docs.n_docs + 1, 0u);
                for (u32 d = 0; d < docs.n_docs; d++)
                    hs[d + 1] = hs[d] + ceil_div_u32(ceil_div_u32(docs.h_doc_off[d + 1] - docs.h_doc_off[d], RS_TILE), RS_GROUP);
                seg.n_groups = hs[docs.n_docs];
                u32 *d_group_doc = ar.alloc<u32>(seg.n_groups);
                seg.group_doc = d_group_doc;
                hs.resize((size_t)docs.n_docs + 1 + seg.n_groups);
                for (u32 d = 0; d < docs.n_docs; d++)
                    for (u32 g = hs[d]; g < hs[d + 1]; g++) hs[(size_t)docs.n_docs + 1 + g] = d;
                // (the documents of many groups, for the spine: see RsSeg)
                const size_t big_at = hs.size();
                for (u32 d = 0; d < docs.n_docs; d++)
                    if (hs[d + 1] - hs[d] > (u32)RS_SPINE_DOC_GROUPS) hs.push_back(d);
                seg.n_big = (u32)(hs.size() - big_at);
                HIP_CHECK(hipMemcpyAsync(d_group0, hs.data(), ((size_t)docs.n_docs + 1) * 4, hipMemcpyHostToDevice, ctx.stream));
                HIP_CHECK(hipMemcpyAsync(d_group_doc, hs.data() + docs.n_docs + 1, (size_t)seg.n_groups * 4, hipMemcpyHostToDevice, ctx.stream));
                if (seg.n_big) {
                    u32 *d_big = ar.alloc<u32>(seg.n_big);
                    seg.big_docs = d_big;
                    HIP_CHECK(hipMemcpyAsync(d_big, hs.data() + big_at, (size_t)seg.n_big * 4, hipMemcpyHostToDevice, ctx.stream));
                }
            }
            docs.bits = 0;                              // (no document number in the keys)
        }
    }
    const size_t mark_lvl = ar.mark();                  // (what a repeated level 0 gives back: not the tables above)
    ctx.did_seg = docs.seg.n_docs != 0;
    if (ctx.stats) ctx.stats->seg_sort = ctx.did_seg;
    if (docs.bits + 3 * bt > 64) return false;          // (no room for a window next to the document number)
    int w = lvl0_window(multi ? longest : n, bt, term_first, docs.bits);
    // (natural-language text over a large alphabet -- upper-case prose with digits and accents: 7 bits a symbol, 3 symbols
    // in a 32-bit key -- resolves far less per symbol than the estimate above assumes; where most suffixes stay tied
    // behind it, the widest window that fits 64 bits costs less than the rounds it saves: real prose 7.45 -> 6.55 ms)
    // Which of the two it is comes from the text itself ("The plan of a first build"): when more than half
    // of the sample's suffixes share the symbols of the narrow window with another suffix of the SAMPLE -- 8192
    // consecutive suffixes --, most of the corpus is tied behind it.  (A speculative build does as the build before.)
    const int w_wide = std::min(12, (64 - docs.bits) / bt);
    bool wide = false;
    if (ctx.plan_wide >= 0) wide = ctx.plan_wide != 0;
    else if (ctx.sample_n && w < w_wide && w * bt + docs.bits <= 32)
        wide = (u64)ctx.sample_dup2[std::min(w, 8)] * 2u > ctx.sample_n;
    if (getenv("EAST_HIP_WIDE")) wide = atoi(getenv("EAST_HIP_WIDE")) != 0;      // (experiments)
    // Variable-length code words in a 32-bit key (ht_code.h): taken where they put at least one symbol more into the key
    // than the fixed width does -- text over a large alphabet in which a few symbols make up most of it (prose: 7 bits a
    // symbol fixed, under 5 coded).  They go before the wide window: the narrow sort is half the passes on two thirds of
    // the bytes, and with 5-6 symbols in the key the rounds have no more to do than behind the 8 symbols of the wide one.
    // They are worth their decoding where the text pays far more bits per symbol than it needs AND the wide window is
    // the plan anyway: the same 64-bit pairs, sorted on fewer bits (48 stream bits: a pass less), holding half as many
    // symbols again.  (Behind a narrow 32-bit key they were measured and lose to the wide fixed-width window: 6.75 against
    // 6.1 ms on 64 x 1 MiB of prose -- twice the suffixes in the rounds.  Forced -- mode 1 -- they take the key width the
    // fixed-width plan would have taken: the tests go through both.)
    bool use_ht = false, ht_wide = false;
    if (ctx.ht_max_len > 0 && !ctx.dry && ctx.knobs.ht_mode != 0) {
        if (ctx.knobs.ht_mode == 1) { use_ht = true; ht_wide = wide || ctx.knobs.force_wide_keys || w * bt + docs.bits > 32; }
        else if (ctx.plan_ht >= 0) { use_ht = ctx.plan_ht != 0; ht_wide = ctx.plan_ht == 2; }
        else if (docs.seg.n_docs) {
            // (segmented sort: all 32 bits of a narrow key are text -- six and a half symbols of prose, sorted in four passes
            // of 8-byte pairs; measured on 64 x 1 MiB of prose: 5.29 ms, against 5.35 with 48 stream bits in 64-bit keys,
            // 5.74 with the wide fixed-width window and 6.0 with the narrow one)
            use_ht = ctx.ht_mean_len <= (double)bt - 1.5;
            ht_wide = false;
            if (use_ht) wide = false;
        } else { use_ht = ht_wide = wide && ctx.ht_mean_len <= (double)bt - 1.5; }
        if (use_ht && !ht_wide && (32 - docs.bits < 20 || ctx.knobs.force_wide_keys)) ht_wide = true;
        if (use_ht && ht_wide && 64 - docs.bits < 20) use_ht = false;
    }
    int ht_sb = !use_ht ? 0 : ht_wide ? std::min(HT_MAX_STREAM, 64 - docs.bits) : std::min(HT_MAX_STREAM, 32 - docs.bits);
    // (a wide key is cut to whole radix digits: 42 stream bits next to 6 document bits sort in five global passes where
    // 48 take six -- measured on 64 x 1 MiB of prose: 5.77 against 5.82 ms, the 8.6 symbols of the shorter key are enough)
    if (use_ht && ht_wide) ht_sb -= (ht_sb + docs.bits) % RS_DB;
    // (experiments: EAST_HIP_HT_SB=<bits> overrides the stream bits of a wide key -- 42 next to 6 document bits is a pass less than 48)
    if (use_ht && ht_wide && getenv("EAST_HIP_HT_SB")) ht_sb = std::max(20, std::min(ht_sb, atoi(getenv("EAST_HIP_HT_SB"))));
    ctx.did_ht = use_ht ? (ht_wide ? 2 : 1) : 0;
    if (ctx.stats) ctx.stats->ht_keys = ctx.did_ht;
    if (wide) w = std::max(w, w_wide);
    ctx.did_wide = wide;
    // (experiments, DESIGN.md 5.2: EAST_HIP_WINDOW=<symbols> overrides the width of the first window)
    if (getenv("EAST_HIP_WINDOW")) w = std::max(3, std::min(atoi(getenv("EAST_HIP_WINDOW")), std::min(12, (64 - docs.bits) / bt)));
    u32 n_names = 0;
    const HtKeys hk{ctx.ht_enc, ctx.ht_dec, ctx.ht_max_len, ht_sb};
    auto level0 = [&](bool allow_fused) {
        if (use_ht && ht_wide)
            return dc3_level0_bytes<u64>(ctx, s8, 0, n, 0, bt, term_first, sa_out, nullptr, n_names, lcp_out, lcp_capped, docs,
                                         kg_mark, allow_fused, multi ? longest : n, &hk);
        if (use_ht)
            return dc3_level0_bytes<u32>(ctx, s8, 0, n, 0, bt, term_first, sa_out, nullptr, n_names, lcp_out, lcp_capped, docs,
                                         kg_mark, allow_fused, multi ? longest : n, &hk);
        return w * bt + docs.bits <= 32 && !ctx.knobs.force_wide_keys
                   ? dc3_level0_bytes<u32>(ctx, s8, 0, n, w, bt, term_first, sa_out, nullptr, n_names, lcp_out, lcp_capped, docs,
                                           kg_mark, allow_fused, multi ? longest : n)
                   : dc3_level0_bytes<u64>(ctx, s8, 0, n, w, bt, term_first, sa_out, nullptr, n_names, lcp_out, lcp_capped, docs,
                                           kg_mark, allow_fused, multi ? longest : n);
    };
    bool ok;
    const Stats stats_in = ctx.stats ? *ctx.stats : Stats();
    try {
        ok = level0(true);
    } catch (const FusedAbort &) {
        if (g_trace) fprintf(stderr, "[east_hip] fused finish gave up (a repeat too long to order directly): full sort\n");
        ar.release(mark_lvl);
        if (kg_mark) kg_mark->k = kg_k_in;
        // nothing of the abandoned attempt is left behind: its pass counts (the bench's byte model reads them), and the
        // "k-gram marks incomplete" flag it may have raised -- the full sort's placement pass marks every bucket start
        if (ctx.stats) *ctx.stats = stats_in;
        if (ctx.kg_bad && !ctx.dry) HIP_CHECK(hipMemsetAsync(ctx.kg_bad, 0, sizeof(u32), ctx.stream));
        ok = level0(false);
    }
    ar.release(mark);
    return ok;
}

