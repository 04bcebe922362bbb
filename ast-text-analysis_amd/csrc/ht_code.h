// ht_code.h -- order-preserving variable-length codes for the first-level window keys.
//
// The window sort keys every suffix by its first w symbols at a FIXED number of bits per symbol (window_sort.h).
// Natural-language text over a hundred-odd distinct symbols pays 7 bits for each of them although two dozen letters
// make up nearly all of the text: 3 symbols in a 32-bit key next to a document number, three quarters of the suffixes
// tied behind it.  Here the symbols get an ALPHABETIC prefix code instead -- code words ordered like the symbols
// (optimal: Garsia-Wachs, from the symbol counts of the text itself), so that the bit string of a coded suffix compares
// like the suffix, and so does any prefix of it: the key is the first bits of that string,
//     stream(q) = code(x[q]) . stream(q + 1),   cut to the key width;  all zero behind a terminator's code word,
// and English-like text gets five or six symbols into the bits that held three.
// What changes for the consumers of the keys: a key holds a VARIABLE number of whole symbols (ht_scan counts them by
// walking the code words with a 4096-entry table: top 12 bits of the rest of the stream -> length, symbol, "is the
// terminator"), the LCP of two keys is the number of whole code words inside their common bit prefix, and a tie group
// shares as many symbols as its key holds whole code words -- the refinement rounds carry that number per position
// (window_sort.h: xdep).
#pragma once
#include "common.h"
#include <algorithm>

#define HT_MAX_LEN 12                   // longest code word (bits): the decode table is indexed by 12 bits
#define HT_MIN_LEN 3                    // shortest: a stream of 48 bits then holds at most 16 symbols (HtWindowGen looks 16 ahead)
#define HT_MAX_STREAM 48                // ... which is the most stream bits a key may carry
#define HT_DEC_SIZE (1u << HT_MAX_LEN)

// ---- host: the code -----------------------------------------------------------------------------------------------
// Leaf depths of an optimal alphabetic tree for the weights w (Garsia-Wachs: combine the leftmost pair that is locally
// minimal, move the new node left past everything lighter; the depths of the resulting tree are those of an optimal
// ALPHABETIC tree).  n >= 2.
static std::vector<int> ht_garsia_wachs(const std::vector<u64> &w)
{
    const int n = (int)w.size();
    struct Node { u64 w; int l, r; };
    std::vector<Node> nodes;
    nodes.reserve(2 * n);
    std::vector<int> seq(n);
    for (int i = 0; i < n; i++) { nodes.push_back(Node{w[i], -1, -1}); seq[i] = i; }
    while (seq.size() > 1) {
        size_t k = 1;
        while (k + 1 < seq.size() && nodes[seq[k - 1]].w > nodes[seq[k + 1]].w) k++;
        const Node x{nodes[seq[k - 1]].w + nodes[seq[k]].w, seq[k - 1], seq[k]};
        const int id = (int)nodes.size();
        nodes.push_back(x);
        seq.erase(seq.begin() + (k - 1), seq.begin() + (k + 1));
        size_t pos = k - 1;
        while (pos > 0 && nodes[seq[pos - 1]].w < x.w) pos--;
        seq.insert(seq.begin() + pos, id);
    }
    std::vector<int> depth(nodes.size(), 0), len(n, 0);
    std::vector<int> stack{seq[0]};
    while (!stack.empty()) {
        const int v = stack.back();
        stack.pop_back();
        if (nodes[v].l < 0) { len[v] = depth[v]; continue; }
        depth[nodes[v].l] = depth[nodes[v].r] = depth[v] + 1;
        stack.push_back(nodes[v].l);
        stack.push_back(nodes[v].r);
    }
    return len;
}

// Code words for symbols 0 .. n-1 in their order (weights w, all >= 1): lengths in [HT_MIN_LEN, HT_MAX_LEN], codes
// strictly increasing as left-aligned bit strings and prefix-free.  code[i] holds the len[i] bits right-aligned.
// Returns false if no such code was found (the caller keeps fixed-width keys).
static bool ht_build_code(const std::vector<u64> &w_in, std::vector<u32> &code, std::vector<int> &len)
{
    const int n = (int)w_in.size();
    if (n < (1 << HT_MIN_LEN) || n > 256) return false;   // (fewer symbols cannot fill the code space with words this long)
    std::vector<u64> w = w_in;
    u64 total = 0;
    for (u64 x : w) total += x;
    for (int attempt = 0; attempt < 24; attempt++) {
        len = ht_garsia_wachs(w);
        const int longest = *std::max_element(len.begin(), len.end()), shortest = *std::min_element(len.begin(), len.end());
        if (longest <= HT_MAX_LEN && shortest >= HT_MIN_LEN) break;
        if (longest > HT_MAX_LEN) {
            // too deep: every symbol gets a floor of its weight (the rare ones move up), doubled until the tree fits
            const u64 floor_w = std::max<u64>(1, (total >> (HT_MAX_LEN - 2)) << attempt);
            for (int i = 0; i < n; i++) w[i] = std::max(w[i], floor_w);
        }
        if (shortest < HT_MIN_LEN) {
            // one symbol carries nearly half of the text: its weight is capped
            u64 t2 = 0;
            for (u64 x : w) t2 += x;
            for (int i = 0; i < n; i++) w[i] = std::min(w[i], std::max<u64>(1, t2 / 10));
        }
        if (attempt == 23) return false;
    }
    // canonical alphabetic assignment: the next code word starts at the smallest multiple of its own size not below
    // the end of the one before (in units of 2^-HT_MAX_LEN)
    code.assign(n, 0);
    u32 c = 0;
    for (int i = 0; i < n; i++) {
        if (len[i] < HT_MIN_LEN || len[i] > HT_MAX_LEN) return false;
        const u32 unit = 1u << (HT_MAX_LEN - len[i]);
        c = (c + unit - 1) & ~(unit - 1);
        if (c + unit > HT_DEC_SIZE) return false;
        code[i] = c >> (HT_MAX_LEN - len[i]);
        c += unit;
    }
    return true;
}

// The two device tables of a code over the byte stream's symbols (1 .. sigma_t text codes in order, then the terminator
// class 0xFF as the largest symbol):
//   enc[byte]  = code << 8 | len                                   (256 entries; unused bytes: 0)
//   dec[top 12 bits of a stream] = sym << 8 | is_term << 7 | len   (sym: the byte value; holes of the code space: len 0)
static bool ht_make_tables(const std::vector<u64> &byte_counts, u32 sigma_t, u64 n_strings, std::vector<u32> &enc,
                           std::vector<uint16_t> &dec, double *mean_len = nullptr)
{
    std::vector<u64> w(sigma_t + 1);
    for (u32 c = 1; c <= sigma_t; c++) w[c - 1] = byte_counts[c] + 1;
    w[sigma_t] = n_strings + 1;                         // the terminator class
    std::vector<u32> code;
    std::vector<int> len;
    if (!ht_build_code(w, code, len)) return false;
    enc.assign(256, HT_MIN_LEN);                        // (bytes that never occur: a harmless code word)
    dec.assign(HT_DEC_SIZE, 0);
    double bits = 0, tot = 0;
    for (u32 i = 0; i <= sigma_t; i++) {
        const u32 byte = i < sigma_t ? i + 1 : 0xFFu;
        enc[byte] = (code[i] << 8) | (u32)len[i];
        const u32 first = code[i] << (HT_MAX_LEN - len[i]), count = 1u << (HT_MAX_LEN - len[i]);
        for (u32 x = 0; x < count; x++) dec[first + x] = (uint16_t)((byte << 8) | (i == sigma_t ? 0x80u : 0u) | (u32)len[i]);
        bits += (double)w[i] * len[i];
        tot += (double)w[i];
    }
    if (mean_len) *mean_len = bits / tot;
    return true;
}

// ---- device: reading a key ----------------------------------------------------------------------------------------
struct HtScan {
    u32 whole;      // whole code words of text in front of the first terminator / the cut of the key
    u32 common;     // ... of them inside the first `cb` bits (the LCP with a key that shares exactly cb bits)
    u32 top;        // ... of them inside the first `tb` bits
    bool term;      // a whole terminator code word follows those `whole` symbols (the key is unique)
    bool term_top;  // ... and it ends inside the first `tb` bits
};

// stream: the coded suffix, left-aligned in S (u32 / u64), sb of its bits valid; dec: the decode table (LDS or global;
// entries of 8 or 16 bits: length in bits 0-6, "terminator" in bit 7)
template <class S, class Table> __device__ __forceinline__ HtScan ht_scan(S stream, int sb, const Table &dec, int cb, int tb = 0)
{
    HtScan r{0u, 0u, 0u, false, false};
    int pos = 0;
    while (pos < sb) {
        const u32 e = dec[(u32)((S)(stream << pos) >> (sizeof(S) * 8 - HT_MAX_LEN))];
        const int len = (int)(e & 0x7Fu);
        if (len == 0 || pos + len > sb) break;          // (a hole of the code space only turns up in the zero padding)
        if (e & 0x80u) { r.term = true; r.term_top = pos + len <= tb; break; }
        r.whole++;
        pos += len;
        if (pos <= cb) r.common++;
        if (pos <= tb) r.top++;
    }
    return r;
}

// Reading the keys of CONSECUTIVE ranks: neighbours in the sorted order share most of their bits, and with the bits the
// code word boundaries inside them.  The walk keeps the boundaries of the current key as a bit mask (bit p - 1: a text
// code word ends behind bit p of the stream); moving on to the next key keeps what lies inside the common bits and goes on
// decoding from there -- and only as far as somebody needs: the LCP entry of a rank wants the boundaries inside the bits it
// shares with its neighbours, the whole key is read only where a tie group or a hot bucket is at stake.
template <class S> struct HtWalk {
    u64 bounds = 0;
    int pos = 0;            // the last boundary found so far (bits decoded)
    int term_end = 0;       // where the terminator's code word ends (0: none met)
    bool done = false;      // the rest of the key holds no further whole code word

    static __device__ __forceinline__ u64 upto(int bits) { return bits >= 64 ? ~0ull : ((u64)1 << bits) - 1ull; }
    // the walk of a key that shares its first `cb` bits with the key walked so far
    __device__ __forceinline__ void inherit(int cb)
    {
        bounds &= upto(cb);
        if (term_end > cb) term_end = 0;
        pos = bounds ? 64 - __clzll((long long)bounds) : 0;
        done = term_end != 0;
    }
    // decode on until every boundary inside the first `need` bits is known
    template <class Table> __device__ __forceinline__ void extend(S stream, int sb, const Table &dec, int need)
    {
        while (!done && pos < need) {
            const u32 e = dec[(u32)((S)(stream << pos) >> (sizeof(S) * 8 - HT_MAX_LEN))];
            const int len = (int)(e & 0x7Fu);
            if (len == 0 || pos + len > sb) { done = true; break; }
            if (e & 0x80u) { term_end = pos + len; done = true; break; }
            pos += len;
            bounds |= (u64)1 << (pos - 1);
        }
    }
    __device__ __forceinline__ u32 symbols_within(int bits) const { return (u32)__popcll(bounds & upto(bits)); }
    // (valid once extend() ran with need = sb)
    __device__ __forceinline__ HtScan result(int cb, int tb) const
    {
        return HtScan{(u32)__popcll(bounds), symbols_within(cb), symbols_within(tb), term_end != 0, term_end != 0 && term_end <= tb};
    }
};
