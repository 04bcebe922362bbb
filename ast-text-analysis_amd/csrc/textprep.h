// textprep.h -- text preparation on the device: raw UTF-8 bytes of all documents ->
// the EASA symbol stream east_hip_build consumes.
//
// Replaces, for a whole collection at once, the host-side chain in front of the hot
// path (reference east/utils.py:31-79 and east/asts/utils.py:25-40):
//   prepare_text      utf-8 decode (errors='replace') + upper()                 utils.py:31-34
//   tokenize          re.findall("[\w']+", re.U)                                utils.py:37-38
//   text_to_strings_collection   keep tokens with len > 2 and not isdigit(),
//                     concatenate groups of 3, empty -> [" "]                   utils.py:49-79
//   make_unique_endings + "".join   string i followed by U+0A00+i               asts/utils.py:25-40
//
// The Unicode knowledge stays in Python: the caller passes, for every code point below
// U+0A00 (the method's text domain), its class (bit 0: matches [\w'], bit 1: str.isdigit)
// and its 1:1 upper-case mapping; for the code points from U+0A00 up, bitmaps of the word
// and digit characters and the (few hundred) 1:1 upper mappings.  A word character beyond
// U+0A00 that survives the token filter is outside the method's domain and raises.
// Input layout: the texts concatenated, every text followed by one 0xFF byte (never valid
// UTF-8: it decodes to a U+FFFD separator and ends a truncated sequence exactly like the
// end of the data does).
//
// Everything is flags + prefix sums + scatters, all streaming:
//   bytes  --is_start / scan-->  code points  --class-->  token starts/ends  --scan-->
//   tokens --keep / scans-->  per-document counts  -->  symbols
#pragma once
#include "common.h"
#include "scan.h"

#define TP_TEXT_LIMIT EAST_HIP_TERMINATOR_START      // 0x0A00
#define TP_REPLACEMENT 0xFFFDu
#define TP_CLASS_WORD 1u
#define TP_CLASS_DIGIT 2u

// ---- UTF-8, errors='replace' (CPython: every maximal ill-formed subpart -> one U+FFFD) ----
// Length of the unit that starts at byte i and its code point (U+FFFD if ill-formed).
__device__ __forceinline__ u32 tp_unit(const uint8_t *__restrict__ b, u64 i, u64 end, u32 &cp)
{
    const u32 x = b[i];
    cp = TP_REPLACEMENT;
    if (x < 0x80u) { cp = x; return 1; }
    u32 need, lo2 = 0x80u, hi2 = 0xBFu;
    if (x >= 0xC2u && x <= 0xDFu) need = 2;
    else if (x >= 0xE0u && x <= 0xEFu) { need = 3; if (x == 0xE0u) lo2 = 0xA0u; if (x == 0xEDu) hi2 = 0x9Fu; }
    else if (x >= 0xF0u && x <= 0xF4u) { need = 4; if (x == 0xF0u) lo2 = 0x90u; if (x == 0xF4u) hi2 = 0x8Fu; }
    else return 1;                                   // C0, C1, F5..FF, stray continuation
    if (i + 1 >= end) return 1;
    const u32 b1 = b[i + 1];
    if (b1 < lo2 || b1 > hi2) return 1;
    if (need == 2) { cp = ((x & 0x1Fu) << 6) | (b1 & 0x3Fu); return 2; }
    if (i + 2 >= end) return 2;
    const u32 b2 = b[i + 2];
    if (b2 < 0x80u || b2 > 0xBFu) return 2;
    if (need == 3) { cp = ((x & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu); return 3; }
    if (i + 3 >= end) return 3;
    const u32 b3 = b[i + 3];
    if (b3 < 0x80u || b3 > 0xBFu) return 3;
    cp = ((x & 0x07u) << 18) | ((b1 & 0x3Fu) << 12) | ((b2 & 0x3Fu) << 6) | (b3 & 0x3Fu);
    return 4;
}

// A byte starts a unit unless it is a continuation byte consumed by the unit of the nearest
// non-continuation byte at most 3 positions before it (UTF-8 is self-synchronising).
struct TpStartIn {
    const uint8_t *b;
    u32 n;
    __device__ __forceinline__ u32 operator()(u32 i) const
    {
        if (i >= n) return 0u;
        const u32 x = b[i];
        if (x < 0x80u || x > 0xBFu) return 1u;
        for (u32 back = 1; back <= 3 && back <= i; back++) {
            const u32 y = b[i - back];
            if (y < 0x80u || y > 0xBFu) {
                u32 cp;
                return back < tp_unit(b, i - back, n, cp) ? 0u : 1u;
            }
        }
        return 1u;
    }
};

// Unicode data of the caller's interpreter (see include/east_hip.h)
struct TpTables {
    const uint8_t *cp_class;      // [0x0A00] bit 0 word, bit 1 digit
    const u32 *cp_upper;          // [0x0A00] 1:1 upper
    const u32 *word_hi;           // bitmaps over [0x0A00, 0x110000)
    const u32 *digit_hi;
    const u32 *hi_from, *hi_to;   // sorted 1:1 upper mappings of code points >= 0x0A00
    u32 n_hi;
};

// ---- ranks without a per-element index --------------------------------------------------------------------------------
// "The how-manyth code point is this byte", "the how-manyth token is this code point": rounds 2-3 wrote such numbers out
// for EVERY element (4 bytes per byte of text, written once and read two or three times -- the heaviest arrays of the whole
// preparation).  Now only the number of set positions in front of every BLOCK of 256 positions is kept
// (tp_block_counts_kernel + a scan over n / 256 entries); a consumer finds the rest among its own block's positions with a
// ballot or a few shuffles, and the handful of per-document look-ups count the up to 255 positions in front of theirs.
#define TP_RANK_BLOCK 256
template <class Pred>
__global__ __launch_bounds__(BLOCK) void tp_block_counts_kernel(Pred pred, u32 n, u32 n_blocks, u32 *__restrict__ counts)
{
    // 8 positions per thread, 32 threads (half a wavefront) per block of 256; counts[n_blocks] = 0 (the scan's last entry: the total)
    const u32 t = blockIdx.x * BLOCK + threadIdx.x;
    u32 cnt = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const u64 p = (u64)t * 8u + e;
        if (p < n) cnt += pred((u32)p);
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o, WAVE);
    const u32 b = t >> 5;
    if ((threadIdx.x & 31u) == 0 && b <= n_blocks) counts[b] = b < n_blocks ? cnt : 0u;
}

// set positions in [0, pos), from the blocks' prefix sums
// by a whole wavefront (every lane calls it with the same pos and gets the result): the up to 255 positions in front of
// pos inside its block, four per lane.  The per-document look-ups below are a wavefront per document (one thread per
// document walking its 255 positions alone: 0.1 ms per launch for a handful of documents, five launches per call).
template <class Pred> __device__ __forceinline__ u32 tp_rank_before_wave(const Pred &pred, const u32 *__restrict__ prefix, u32 pos)
{
    static_assert(TP_RANK_BLOCK == 4 * WAVE, "four positions per lane");
    const u32 base = pos & ~(u32)(TP_RANK_BLOCK - 1);
    u32 c = 0;
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
        const u32 p = base + k * WAVE + lane_id();
        c += p < pos ? pred(p) : 0u;
    }
    return prefix[pos / TP_RANK_BLOCK] + wave_sum(c);
}

// one position per thread, workgroup = one block of 256: the set positions of the block in front of this thread's
// (lds4: 4 words; a barrier inside -- every thread of the workgroup calls it)
__device__ __forceinline__ u32 tp_rank_in_block(bool set, u32 *lds4)
{
    const u64 bal = __ballot(set);
    if (lane_id() == 0) lds4[wave_id()] = (u32)__popcll(bal);
    __syncthreads();
    u32 before = (u32)__popcll(bal & (((u64)1 << lane_id()) - 1ull));
    for (u32 w = 0; w < wave_id(); w++) before += lds4[w];
    return before;
}

// per start byte: decode, upper, classify; cpu[idx] = code point after upper, cw[idx] = class
// (byte_prefix: unit starts in front of every block of 256 bytes -- the code point index of a start byte is that plus the
// starts in front of it inside its block)
__global__ __launch_bounds__(BLOCK) void tp_decode_kernel(const uint8_t *__restrict__ b, u32 n_bytes,
                                                          const u32 *__restrict__ byte_prefix, TpTables T,
                                                          u32 *__restrict__ cpu, uint8_t *__restrict__ cw)
{
    __shared__ u32 lds4[WAVES_PER_BLOCK];
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    const TpStartIn st{b, n_bytes};
    const bool is_start = i < n_bytes && st(i);
    const u32 idx = byte_prefix[blockIdx.x] + tp_rank_in_block(is_start, lds4);
    if (!is_start) return;
    u32 cp;
    (void)tp_unit(b, i, n_bytes, cp);
    if (cp < TP_TEXT_LIMIT) {
        cp = T.cp_upper[cp];
    } else {                                       // rare: a handful of high code points have a 1:1 upper
        u32 lo = 0, hi = T.n_hi;
        while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (T.hi_from[mid] < cp) lo = mid + 1; else hi = mid; }
        if (lo < T.n_hi && T.hi_from[lo] == cp) cp = T.hi_to[lo];
    }
    u32 cls;
    if (cp < TP_TEXT_LIMIT) {
        cls = T.cp_class[cp];
    } else {                                       // words beyond the domain are refused when (if) they are emitted
        const u32 k = cp - TP_TEXT_LIMIT;
        cls = ((T.word_hi[k >> 5] >> (k & 31u)) & 1u) | (((T.digit_hi[k >> 5] >> (k & 31u)) & 1u) << 1);
    }
    cpu[idx] = cp;
    cw[idx] = (uint8_t)cls;
}

// The same for text in which every byte is a code point of its own (ASCII; stray bytes >= 0x80 each decode to U+FFFD):
// the class of a byte is a 256-entry table (built on the host from the caller's tables: class of the upper-cased code
// point), the code points themselves are not stored at all -- tp_emit_kernel maps the bytes it keeps through a second
// 256-entry table.  Sixteen bytes per thread, one 16-byte load and one 16-byte store (the buffers are padded).
__global__ __launch_bounds__(BLOCK) void tp_classify_bytes_kernel(const uint8_t *__restrict__ b, u32 n_bytes,
                                                                  const uint8_t *__restrict__ cls256,
                                                                  uint8_t *__restrict__ cw)
{
    __shared__ uint8_t cls[256];
    cls[threadIdx.x] = cls256[threadIdx.x];             // (BLOCK == 256)
    __syncthreads();
    const u32 i = (blockIdx.x * BLOCK + threadIdx.x) * 16u;
    if (i >= n_bytes) return;
    uint4 x;                                            // (a chunk of the streamed preparation starts at any byte)
    __builtin_memcpy(&x, b + i, 16);
    auto four = [&](u32 v) -> u32 {
        return (u32)cls[v & 0xFFu] | ((u32)cls[(v >> 8) & 0xFFu] << 8) | ((u32)cls[(v >> 16) & 0xFFu] << 16) | ((u32)cls[v >> 24] << 24);
    };
    *reinterpret_cast<uint4 *>(cw + i) = uint4{four(x.x), four(x.y), four(x.z), four(x.w)};
}

// code-point index of every document's first byte
// (byte_prefix == nullptr: every byte is a code point of its own)
__global__ __launch_bounds__(BLOCK) void tp_doc_cp_offsets_kernel(const uint8_t *__restrict__ b, u32 n_bytes,
                                                                  const u32 *__restrict__ byte_prefix,
                                                                  const u32 *__restrict__ text_off, u32 n_docs,
                                                                  u32 *__restrict__ doc_cp_off)
{
    const u32 d = blockIdx.x * WAVES_PER_BLOCK + wave_id();   // a wavefront per document
    if (d > n_docs) return;
    const u32 v = byte_prefix ? tp_rank_before_wave(TpStartIn{b, n_bytes}, byte_prefix, text_off[d]) : text_off[d];
    if (lane_id() == 0) doc_cp_off[d] = v;
}

struct TpTokStartIn {                            // 1 at the first code point of a token; defined on [0, n]
    const uint8_t *cw;
    u32 n;
    __device__ __forceinline__ u32 operator()(u32 p) const
    {
        return (p < n && (cw[p] & TP_CLASS_WORD) && !(p > 0 && (cw[p - 1] & TP_CLASS_WORD))) ? 1u : 0u;
    }
};

// A word position p belongs to token (token starts in [0, p]) - 1; tok_prefix: the token starts in front of every block
// of 256 code points (TP_RANK_BLOCK): a thread's eight positions lie in one block, the starts in front of them inside
// the block come from the 32 threads of that block (half a wavefront) by shuffles.
// tok_nd[k] (zeroed by the caller) becomes 1 when token k holds a character that is not a digit
#define TP_VEC 8                                  // code points per thread in the two per-code-point passes (8-byte class loads)
__global__ __launch_bounds__(BLOCK) void tp_token_bounds_kernel(const uint8_t *__restrict__ cw,
                                                                const u32 *__restrict__ tok_prefix, u32 n_cp,
                                                                u32 *__restrict__ tstart, u32 *__restrict__ tend,
                                                                u32 *__restrict__ tok_nd)
{
    // (cw is allocated with TP_VEC bytes of padding behind n_cp and starts 16-byte aligned)
    const u32 p0 = (blockIdx.x * BLOCK + threadIdx.x) * TP_VEC;
    const bool live = p0 < n_cp;
    const u64 c8 = live ? *reinterpret_cast<const u64 *>(cw + p0) : 0ull;
    u32 prev = live && p0 > 0 ? cw[p0 - 1] : 0u;
    const u32 after = live && p0 + TP_VEC < n_cp ? cw[p0 + TP_VEC] : 0u;
    u32 starts = 0;                                // token starts among the thread's positions
    {
        u32 pv = prev;
#pragma unroll
        for (int e = 0; e < TP_VEC; e++) {
            const u32 c = (u32)(c8 >> (8 * e)) & 0xFFu;
            if (p0 + e < n_cp && (c & TP_CLASS_WORD) && !(pv & TP_CLASS_WORD)) starts++;
            pv = c;
        }
    }
    u32 inc = starts;                              // inclusive prefix over the lanes of the same half-wavefront (= block of 256)
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const u32 y = __shfl_up(inc, o, WAVE);
        if ((lane_id() & 31u) >= (u32)o) inc += y;
    }
    if (!live || !(c8 & 0x0101010101010101ull * TP_CLASS_WORD)) return;          // no word character among the eight
    u32 seen = tok_prefix[p0 / TP_RANK_BLOCK] + inc - starts;          // token starts in front of position p0
#pragma unroll
    for (int e = 0; e < TP_VEC; e++) {
        const u32 p = p0 + e;
        const u32 c = (u32)(c8 >> (8 * e)) & 0xFFu;
        const u32 next = e + 1 < TP_VEC ? (u32)(c8 >> (8 * (e + 1))) & 0xFFu : after;
        if (p < n_cp && (c & TP_CLASS_WORD)) {
            const bool first = !(prev & TP_CLASS_WORD);
            if (first) seen++;
            const u32 k = seen - 1u;
            if (first) tstart[k] = p;
            if (!(p + 1 < n_cp && (next & TP_CLASS_WORD))) tend[k] = p;
            // (ordinary words: the first letter says it; a token that starts with digits hears it from its first other character)
            if (!(c & TP_CLASS_DIGIT) && (first || (prev & TP_CLASS_DIGIT))) tok_nd[k] = 1u;
        }
        prev = c;
    }
}

// keep[k] = len > 2 and not all digits (utils.py:63); klen[k] = kept length or 0
// (n_tok_dev != nullptr: the number of tokens is still on the device -- the streamed preparation launches over an upper bound)
__global__ __launch_bounds__(BLOCK) void tp_token_keep_kernel(const u32 *__restrict__ tstart,
                                                              const u32 *__restrict__ tend,
                                                              const u32 *__restrict__ tok_nd, u32 n_tok,
                                                              u32 *__restrict__ keep, u32 *__restrict__ klen,
                                                              const u32 *__restrict__ n_tok_dev = nullptr)
{
    const u32 k = blockIdx.x * BLOCK + threadIdx.x;
    if (n_tok_dev) n_tok = *n_tok_dev;
    if (k >= n_tok) return;
    const u32 a = tstart[k], e = tend[k];
    const u32 len = e - a + 1u;
    const bool kp = len > 2u && tok_nd[k] != 0u;
    keep[k] = kp ? 1u : 0u;
    klen[k] = kp ? len : 0u;
}

// what the kernels over a chunk's tokens want zeroed: tok_nd[0 .. n_tok] (raised by the token-bounds pass) and the entry
// behind the last token of keep / klen (the scans read one element more than there are tokens) -- n_tok from the device
__global__ __launch_bounds__(BLOCK) void tp_zero_tokens_kernel(u32 *__restrict__ tok_nd, u32 *__restrict__ keep, u32 *__restrict__ klen,
                                                               u32 ub, const u32 *__restrict__ n_tok_dev)
{
    const u32 n_tok = min(*n_tok_dev, ub);
    const u32 i = (blockIdx.x * BLOCK + threadIdx.x) * 4u;
    if (i > n_tok) return;
    if (i + 3u <= ub) *reinterpret_cast<uint4 *>(tok_nd + i) = uint4{0u, 0u, 0u, 0u};     // (the arrays hold ub + 1 words)
    else
        for (u32 k = i; k <= ub; k++) tok_nd[k] = 0u;
    if (i == 0) { keep[n_tok] = 0u; klen[n_tok] = 0u; }
}

// per document: first token, kept tokens, strings m_d, symbols n_d
__global__ __launch_bounds__(BLOCK) void tp_doc_counts_kernel(const u32 *__restrict__ doc_cp_off,
                                                              const uint8_t *__restrict__ cw, u32 n_cp,
                                                              const u32 *__restrict__ tok_prefix,
                                                              const u32 *__restrict__ keep_ex,
                                                              const u32 *__restrict__ klen_ex, u32 n_docs,
                                                              u32 *__restrict__ first_tok, u32 *__restrict__ m_d,
                                                              u32 *__restrict__ n_d)
{
    const u32 d = blockIdx.x * WAVES_PER_BLOCK + wave_id();   // a wavefront per document
    if (d > n_docs) return;
    const TpTokStartIn ts{cw, n_cp};
    const u32 ft = tp_rank_before_wave(ts, tok_prefix, doc_cp_off[d]);      // tokens that start before the document
    const u32 ft1 = d < n_docs ? tp_rank_before_wave(ts, tok_prefix, doc_cp_off[d + 1]) : 0u;
    if (lane_id() != 0) return;
    first_tok[d] = ft;
    if (d == n_docs) return;
    const u32 kd = keep_ex[ft1] - keep_ex[ft];
    const u32 chars = klen_ex[ft1] - klen_ex[ft];
    m_d[d] = kd ? (kd + 2u) / 3u : 1u;                       // utils.py:76-77: an empty collection becomes [" "]
    n_d[d] = kd ? chars + (kd + 2u) / 3u : 2u;
}

__device__ __forceinline__ u32 tp_doc_of_cp(const u32 *__restrict__ doc_cp_off, u32 n_docs, u32 p)
{
    u32 lo = 0, hi = n_docs;          // doc_cp_off[lo] <= p < doc_cp_off[hi]
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (doc_cp_off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

// per kept token: where its first symbol goes, and the terminator that follows its last one (0 = none) --
// the document look-up and the group arithmetic happen once per token, not once per code point
#define TP_DROPPED 0xFFFFFFFFu
__global__ __launch_bounds__(BLOCK) void tp_token_out_kernel(const u32 *__restrict__ tstart,
                                                             const u32 *__restrict__ keep_ex,
                                                             const u32 *__restrict__ klen_ex,
                                                             const u32 *__restrict__ doc_cp_off,
                                                             const u32 *__restrict__ first_tok,
                                                             const u32 *__restrict__ doc_sym_off, u32 n_docs, u32 n_tok,
                                                             const u32 *__restrict__ tend,
                                                             u32 *__restrict__ tok_out, u32 *__restrict__ tok_term,
                                                             uint4 *__restrict__ tok_rec)
{
    // tok_rec[k] = {where the token's first symbol goes, its first code point, its terminator, its last code point}:
    // everything tp_emit_kernel needs about a token in ONE 16-byte load
    const u32 k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n_tok) return;
    if (keep_ex[k + 1u] == keep_ex[k]) { tok_out[k] = TP_DROPPED; tok_term[k] = 0; tok_rec[k] = uint4{TP_DROPPED, 0u, 0u, 0u}; return; }
    const u32 d = tp_doc_of_cp(doc_cp_off, n_docs, tstart[k]);
    const u32 ft = first_tok[d], ft1 = first_tok[d + 1];
    const u32 kidx = keep_ex[k] - keep_ex[ft];               // index among the document's kept tokens
    const u32 kd = keep_ex[ft1] - keep_ex[ft];
    const u32 g = kidx / 3u;
    const u32 out = doc_sym_off[d] + (klen_ex[k] - klen_ex[ft]) + g;
    const u32 term = (kidx % 3u == 2u || kidx + 1u == kd) ? TP_TEXT_LIMIT + g : 0u;
    tok_out[k] = out;
    tok_term[k] = term;
    tok_rec[k] = uint4{out, tstart[k], term, tend[k]};
}

// every code point of a kept token goes to its place; the last one of a group writes the terminator.
// One code point per thread (neighbouring lanes write neighbouring symbols); per code point one class byte, the token's
// number and ONE 16-byte token record.  cpu == nullptr: every byte is a code point of its own -- the code point is
// up256[byte].  (Eight consecutive code points per thread were measured: 0.71 against 0.45 ms -- the stores stride.)
// (the token of a code point: the token starts in front of its block of 256 -- tok_prefix -- plus those of the block up
// to the code point itself, by ballot)
__global__ __launch_bounds__(BLOCK) void tp_emit_kernel(const u32 *__restrict__ cpu, const uint8_t *__restrict__ bytes,
                                                        const u32 *__restrict__ up256, const uint8_t *__restrict__ cw,
                                                        const u32 *__restrict__ tok_prefix, const uint4 *__restrict__ tok_rec,
                                                        u32 n_cp, u32 *__restrict__ sym, u32 *__restrict__ high)
{
    __shared__ u32 up[256];
    __shared__ u32 lds4[WAVES_PER_BLOCK];
    if (!cpu) up[threadIdx.x] = up256[threadIdx.x];    // (BLOCK == 256)
    const u32 p = blockIdx.x * BLOCK + threadIdx.x;
    // (everything that does not depend on the token record is requested here, unconditionally -- a position behind the end reads
    // the last one --: the code point / byte used to be fetched behind the record, a third round trip)
    const u32 pc = p < n_cp ? p : n_cp - 1u;
    const u32 c_raw = cw[pc], c_prev = cw[pc > 0 ? pc - 1u : 0u];
    const u32 cp_raw = cpu ? cpu[pc] : (u32)bytes[pc];
    const u32 tok0 = tok_prefix[blockIdx.x];
    const u32 c = p < n_cp ? c_raw : 0u;
    const bool word = (c & TP_CLASS_WORD) != 0;
    const bool start = word && !(p > 0 && (c_prev & TP_CLASS_WORD));
    const u32 before = tp_rank_in_block(start, lds4);   // (its barrier also covers up[])
    if (!word) return;
    const uint4 rec = tok_rec[tok0 + before + (start ? 1u : 0u) - 1u];
    if (rec.x == TP_DROPPED) return;                        // token dropped
    const u32 out = rec.x + (p - rec.y);
    const u32 cp = cpu ? cp_raw : up[cp_raw];
    if (cp >= TP_TEXT_LIMIT) *high = 1u;                    // kept text at or above U+0A00: the build takes the tagged encoding
    sym[out] = cp;
    if (rec.z && p == rec.w) sym[out + 1u] = rec.z;
}

// Kept word characters at or above U+0A00 were found: the terminators are rewritten in the tagged encoding
// (EAST_HIP_TERMINATOR_TAG | index), in which text may be any code point.
__global__ __launch_bounds__(BLOCK) void tp_tag_terminators_kernel(const u32 *__restrict__ tstart,
                                                                   const u32 *__restrict__ tend,
                                                                   const u32 *__restrict__ tok_out,
                                                                   const u32 *__restrict__ tok_term, u32 n_tok,
                                                                   u32 *__restrict__ sym)
{
    const u32 k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n_tok || tok_out[k] == TP_DROPPED || !tok_term[k]) return;
    sym[tok_out[k] + (tend[k] - tstart[k]) + 1u] = EAST_HIP_TERMINATOR_TAG | (tok_term[k] - TP_TEXT_LIMIT);
}

__global__ __launch_bounds__(BLOCK) void tp_tag_empty_docs_kernel(const u32 *__restrict__ first_tok,
                                                                  const u32 *__restrict__ keep_ex,
                                                                  const u32 *__restrict__ doc_sym_off, u32 n_docs,
                                                                  u32 *__restrict__ sym)
{
    const u32 d = blockIdx.x * BLOCK + threadIdx.x;
    if (d < n_docs && keep_ex[first_tok[d + 1]] == keep_ex[first_tok[d]]) sym[doc_sym_off[d] + 1u] = EAST_HIP_TERMINATOR_TAG;
}

__global__ __launch_bounds__(BLOCK) void tp_empty_docs_kernel(const u32 *__restrict__ first_tok,
                                                              const u32 *__restrict__ keep_ex,
                                                              const u32 *__restrict__ doc_sym_off, u32 n_docs,
                                                              u32 *__restrict__ sym)
{
    const u32 d = blockIdx.x * BLOCK + threadIdx.x;
    if (d >= n_docs) return;
    if (keep_ex[first_tok[d + 1]] == keep_ex[first_tok[d]]) {
        sym[doc_sym_off[d]] = 32u;                          // [" "]
        sym[doc_sym_off[d] + 1u] = TP_TEXT_LIMIT;
    }
}

// ---- the streamed preparation: a chunk of the byte stream at a time --------------------------------------------------
// The raw text reaches the device in chunks (a copy stream, fed by a thread of its own: copies out of pageable memory block
// their caller) and every chunk is prepared while the next one is on its way.  The host cuts the stream where no token
// and no UTF-8 unit can span the cut -- behind a document's separator, or behind an ASCII byte that is no word character --,
// so a chunk is prepared exactly like a small collection of its own: the kernels above on chunk-local code point and token
// numbers.  What crosses a cut is carried on the device: where the document under way starts in the symbol stream, and
// how many kept tokens and symbols of it earlier chunks have emitted (TpCarry; two slots used in turn) -- they decide
// which of a chunk's tokens close a string of three and where its symbols go.  Nothing comes back to the host before the end.
struct TpCarry {
    u32 sym_base;       // first symbol of the document the next chunk starts in (or continues)
    u32 kept, chars;    // kept tokens / symbols (without terminators) of that document emitted so far
    u32 pad;
};

// per local document i of the chunk (i <= n_docs: one entry more for the bounds): its first token, the kept tokens and
// symbols of the WHOLE document so far, and -- if the document ends in this chunk -- its length in the symbol stream
__global__ __launch_bounds__(BLOCK) void tp_stream_docs_kernel(const u32 *__restrict__ doc_cp_off,
                                                               const uint8_t *__restrict__ cw, u32 n_cp,
                                                               const u32 *__restrict__ tok_prefix,
                                                               const u32 *__restrict__ keep_ex,
                                                               const u32 *__restrict__ klen_ex, u32 n_docs, u32 cont_in,
                                                               u32 cont_out, const TpCarry *__restrict__ carry,
                                                               u32 *__restrict__ first_tok, u32 *__restrict__ n_loc,
                                                               u32 *__restrict__ kept_tot, u32 *__restrict__ chars_tot)
{
    const u32 i = blockIdx.x * WAVES_PER_BLOCK + wave_id();   // a wavefront per document
    if (i > n_docs) return;
    const TpTokStartIn ts{cw, n_cp};
    const u32 ft = tp_rank_before_wave(ts, tok_prefix, doc_cp_off[i]);
    const u32 ft1 = i < n_docs ? tp_rank_before_wave(ts, tok_prefix, doc_cp_off[i + 1]) : 0u;
    if (lane_id() != 0) return;
    first_tok[i] = ft;
    if (i == n_docs) { n_loc[i] = 0; return; }
    u32 kd = keep_ex[ft1] - keep_ex[ft], ch = klen_ex[ft1] - klen_ex[ft];
    if (i == 0 && cont_in) { kd += carry->kept; ch += carry->chars; }
    kept_tot[i] = kd;
    chars_tot[i] = ch;
    const bool complete = !(i + 1 == n_docs && cont_out);
    n_loc[i] = complete ? (kd ? ch + (kd + 2u) / 3u : 2u) : 0u;    // utils.py:76-77: an empty collection becomes [" "]
}

// tp_token_out_kernel for a chunk (off_loc: exclusive scan of n_loc)
__global__ __launch_bounds__(BLOCK) void tp_stream_token_out_kernel(const u32 *__restrict__ tstart, const u32 *__restrict__ tend,
                                                                    const u32 *__restrict__ keep_ex,
                                                                    const u32 *__restrict__ klen_ex,
                                                                    const u32 *__restrict__ doc_cp_off,
                                                                    const u32 *__restrict__ first_tok,
                                                                    const u32 *__restrict__ off_loc,
                                                                    const u32 *__restrict__ kept_tot, u32 n_docs, u32 cont_in,
                                                                    u32 cont_out, const TpCarry *__restrict__ carry,
                                                                    const u32 *__restrict__ n_tok_dev,
                                                                    uint4 *__restrict__ tok_rec)
{
    const u32 k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= *n_tok_dev) return;
    if (keep_ex[k + 1u] == keep_ex[k]) { tok_rec[k] = uint4{TP_DROPPED, 0u, 0u, 0u}; return; }
    const u32 d = tp_doc_of_cp(doc_cp_off, n_docs, tstart[k]);
    const u32 ft = first_tok[d];
    const bool carried = d == 0 && cont_in;
    const u32 kidx = keep_ex[k] - keep_ex[ft] + (carried ? carry->kept : 0u);    // index among the document's kept tokens
    const u32 g = kidx / 3u;
    const u32 out = carry->sym_base + off_loc[d] + (klen_ex[k] - klen_ex[ft]) + (carried ? carry->chars : 0u) + g;
    const bool complete = !(d + 1 == n_docs && cont_out);
    const u32 term = (kidx % 3u == 2u || (complete && kidx + 1u == kept_tot[d])) ? TP_TEXT_LIMIT + g : 0u;
    tok_rec[k] = uint4{out, tstart[k], term, tend[k]};
}

// the documents that end in the chunk: their number of strings, an empty one's [" "], the terminator of a last string
// whose tokens all lie in earlier chunks; every document that starts in the chunk: its offset; and the carry for the next chunk
__global__ __launch_bounds__(BLOCK) void tp_stream_close_docs_kernel(const u32 *__restrict__ off_loc, const u32 *__restrict__ n_loc,
                                                                     const u32 *__restrict__ kept_tot,
                                                                     const u32 *__restrict__ chars_tot, u32 n_docs,
                                                                     u32 doc_first, u32 cont_in, u32 cont_out,
                                                                     const TpCarry *__restrict__ carry,
                                                                     TpCarry *__restrict__ carry_next,
                                                                     u32 *__restrict__ doc_sym_off_all, u32 *__restrict__ m_all,
                                                                     u32 *__restrict__ sym)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_docs) return;
    const u32 base = carry->sym_base + off_loc[i];
    const u32 kd = kept_tot[i];
    const bool carried = i == 0 && cont_in, complete = !(i + 1 == n_docs && cont_out);
    if (!carried) doc_sym_off_all[doc_first + i] = base;
    if (complete) {
        m_all[doc_first + i] = kd ? (kd + 2u) / 3u : 1u;
        if (kd == 0) {
            sym[base] = 32u;                                    // [" "]
            sym[base + 1u] = TP_TEXT_LIMIT;
        } else if (carried && kd == carry->kept && kd % 3u != 0u) {
            // (no kept token of the document in this chunk: the last string's terminator is still to be written)
            sym[base + n_loc[i] - 1u] = TP_TEXT_LIMIT + (kd + 2u) / 3u - 1u;
        }
    }
    if (i + 1 == n_docs) {
        if (complete) *carry_next = TpCarry{base + n_loc[i], 0u, 0u, 0u};
        else *carry_next = TpCarry{base, kd, chars_tot[i], 0u};
    }
}
