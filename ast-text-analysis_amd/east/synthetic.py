# -*- coding: utf-8 -*-
"""BASELINE synthetic corpora: uniform A-Z word streams.

`word_stream_document` returns the text AND the EASA symbol array the product's
own text preparation would produce for it (utils.text_to_strings_collection +
asts.utils.strings_to_symbols), built with numpy so that 64 MiB documents do
not go through Python's regex tokenizer.  tests/test_host_logic.py checks the
two routes give identical symbols.
"""
import numpy as np

TERMINATOR_START = 0x0A00


def _words(rng, n_bytes, lo=3, hi=10):
    n_words = int(n_bytes // ((lo + hi) / 2.0 + 1.0) * 1.1) + 16
    lens = rng.integers(lo, hi + 1, size=n_words).astype(np.int64)
    while int(lens.sum() + n_words) < n_bytes + hi + 1:
        lens = np.concatenate([lens, rng.integers(lo, hi + 1, size=n_words).astype(np.int64)])
    starts = np.cumsum(lens + 1) - (lens + 1)              # word w occupies [start, start+len), then a space
    keep = starts < n_bytes
    lens, starts = lens[keep], starts[keep]
    lens[-1] = min(int(lens[-1]), n_bytes - int(starts[-1]))   # the byte budget may cut the last word
    return lens, starts


def word_stream_document(rng, n_bytes, want_text=True):
    """One document: words of length U{3..10} over A-Z joined by single spaces, cut at n_bytes.

    Returns (text bytes or None, symbols uint32, n_strings): symbols is the
    document's EASA string -- groups of three words (tokens of length <= 2 are
    dropped, utils.py:63) each followed by its terminator 0x0A00+i."""
    lens, starts = _words(rng, n_bytes)
    letters = rng.integers(65, 91, size=int(lens.sum()), dtype=np.uint8)
    text = None
    if want_text:
        buf = np.full(n_bytes, 32, dtype=np.uint8)
        pos = np.repeat(starts - (np.cumsum(lens) - lens), lens) + np.arange(letters.size)
        buf[pos] = letters
        text = buf.tobytes()
    tok = lens > 2                                          # only the cut last word can fail this
    if not tok.all():
        letter_keep = np.repeat(tok, lens)
        letters, lens = letters[letter_keep], lens[tok]
    k = lens.size
    if k == 0:                                              # utils.py:76-77: [" "]
        return text, np.array([32, TERMINATOR_START], dtype=np.uint32), 1
    m = (k + 2) // 3
    tok_idx = np.arange(k, dtype=np.int64)
    out_start = (np.cumsum(lens) - lens) + tok_idx // 3
    is_last = (tok_idx % 3 == 2) | (tok_idx == k - 1)
    term_pos = (out_start + lens)[is_last]
    out = np.empty(int(lens.sum()) + m, dtype=np.uint32)
    mask = np.ones(out.size, dtype=bool)
    mask[term_pos] = False
    out[mask] = letters
    out[term_pos] = np.arange(m, dtype=np.uint32) + np.uint32(TERMINATOR_START)
    return text, out, int(m)


def direct_document(rng, n_symbols):
    """`get_ast([one long string])` mode: n_symbols-1 letters A-Z + the single terminator."""
    out = rng.integers(65, 91, size=n_symbols, dtype=np.uint32)
    out[-1] = TERMINATOR_START
    return out, 1


def keyphrases(rng, doc_symbols, n_keyphrases, doc_offsets=None):
    """1-3 'words': half copied from random corpus positions (deep matches), half fresh random.

    Returns (q_symbols uint32, q_offsets int64) with spaces already removed --
    the layout of east_hip_score_table."""
    parts = []
    n = doc_symbols.size
    for i in range(n_keyphrases):
        length = int(rng.integers(3, 11, size=int(rng.integers(1, 4))).sum())
        if i % 2 == 0 and n > length + 1:
            st = int(rng.integers(0, n - length))
            q = doc_symbols[st:st + length]
            q = q[q < TERMINATOR_START]
            if q.size == 0:
                q = rng.integers(65, 91, size=length, dtype=np.uint32)
        else:
            q = rng.integers(65, 91, size=length, dtype=np.uint32)
        parts.append(q.astype(np.uint32))
    offsets = np.zeros(n_keyphrases + 1, dtype=np.int64)
    np.cumsum([p.size for p in parts], out=offsets[1:])
    return np.concatenate(parts), offsets


def zipf_vocabulary(rng, size=50000, exponent=1.1):
    """Natural-language stand-in for BASELINE config 5 (enwik8 is not available offline): a
    vocabulary of `size` random upper-case words of length 3..12 with Zipf(exponent) frequencies."""
    lens = np.clip(rng.geometric(0.28, size=size) + 2, 3, 12).astype(np.int64)
    letters = rng.integers(65, 91, size=int(lens.sum()), dtype=np.uint8)
    starts = np.cumsum(lens) - lens
    prob = 1.0 / np.arange(1, size + 1, dtype=np.float64) ** exponent
    return {"lens": lens, "starts": starts, "letters": letters, "cdf": np.cumsum(prob / prob.sum())}


def zipf_document(rng, n_bytes, vocab):
    """One document of about n_bytes: Zipf-distributed words joined by single spaces, as the EASA
    symbol array of its 3-word strings (every word has >= 3 letters, so none is filtered).
    Returns (symbols uint32, n_strings)."""
    mean_len = float((vocab["lens"] * np.diff(np.concatenate([[0.0], vocab["cdf"]]))).sum())
    n_words = max(1, int(n_bytes / (mean_len + 1.0)))
    ids = np.searchsorted(vocab["cdf"], rng.random(n_words), side="left").clip(0, vocab["lens"].size - 1)
    lens = vocab["lens"][ids]
    src = np.repeat(vocab["starts"][ids] - (np.cumsum(lens) - lens), lens) + np.arange(int(lens.sum()))
    letters = vocab["letters"][src]
    k = n_words
    m = (k + 2) // 3
    tok_idx = np.arange(k, dtype=np.int64)
    out_start = (np.cumsum(lens) - lens) + tok_idx // 3
    is_last = (tok_idx % 3 == 2) | (tok_idx == k - 1)
    term_pos = (out_start + lens)[is_last]
    out = np.empty(int(lens.sum()) + m, dtype=np.uint32)
    mask = np.ones(out.size, dtype=bool)
    mask[term_pos] = False
    out[mask] = letters
    out[term_pos] = np.arange(m, dtype=np.uint32) + np.uint32(TERMINATOR_START)
    return out, int(m)


# ---- real prose ---------------------------------------------------------------------------------
_PROSE_DIRS = ("/usr/share/perl", "/usr/share/perl5", "/usr/share/doc", "/usr/share/common-licenses", "/usr/lib/python3",
               "/usr/lib/python3.10", "/usr/local/lib/python3.10/dist-packages", "/opt/rocm/share/doc", "/usr/share/vim")
_PROSE_EXT = (".pod", ".rst", ".md", ".txt")
_PROSE_NAMES = ("README", "LICENSE", "COPYING", "NEWS", "copyright", "CHANGELOG", "CHANGES")


def image_prose(max_bytes, keep_duplicates=False):
    """Natural-language text that ships with the container image (POD / reST / Markdown / README /
    licence files), for BASELINE config 5 -- enwik8 is not available offline.  Files with identical
    content are taken once unless keep_duplicates (duplicates = long repeats); code points outside
    the method's domain (>= U+0A00) become spaces.  Returns (bytes, number of files);
    deterministic for a given image (sorted walk)."""
    import hashlib
    import os
    seen, parts, total = set(), [], 0
    for top in _PROSE_DIRS:
        for base, dirs, files in os.walk(top):
            dirs.sort()
            if "/db" in base or "/torch/share" in base:
                continue
            for f in sorted(files):
                if not (f.endswith(_PROSE_EXT) or f.startswith(_PROSE_NAMES)) or "fdb" in f or ".db." in f:
                    continue
                path = os.path.join(base, f)
                try:
                    if not 4096 <= os.path.getsize(path) <= 2000000:
                        continue
                    with open(path, "rb") as fh:
                        data = fh.read()
                except OSError:
                    continue
                text = data.decode("utf-8", errors="replace")
                data = "".join(c if ord(c) < TERMINATOR_START else " " for c in text).encode("utf-8")
                digest = hashlib.sha1(data).digest()
                if digest in seen and not keep_duplicates:
                    continue
                seen.add(digest)
                parts.append(data)
                total += len(data)
                if total >= max_bytes:
                    return b"\n".join(parts)[:max_bytes], len(parts)
    return b"\n".join(parts), len(parts)


# ---- the reference's own benchmark input ----------------------------------------------------------
def worst_case_collection(rng, m, n):
    """The input of the reference's runtime harness (analysis/runtime.py:19-31 on analysis/utils.py:5-9,
    `worst_case_strings_collection(m, n)`): m strings that share one random prefix.  As shipped,
    `utils.random_string(length)` returns length - 2 letters (east/utils.py:86-88), so the "2 differing symbols" are
    empty and the collection is m IDENTICAL strings of n - 4 letters A-Z -- every suffix sits in a tie group of m
    members that only the terminators tell apart, and the longest common prefix is the whole string.
    Returns (symbols uint32 of one document: the m strings, each followed by its terminator; n_strings = m)."""
    length = max(int(n) - 4, 1)
    prefix = rng.integers(65, 91, size=length, dtype=np.uint32)
    out = np.empty((int(m), length + 1), dtype=np.uint32)
    out[:, :length] = prefix
    out[:, length] = np.arange(int(m), dtype=np.uint32) + np.uint32(TERMINATOR_START)
    return out.reshape(-1), int(m)


def repeated_passage_document(rng, passage_symbols, copies):
    """`get_ast([one string])` on a passage of random letters written `copies` times in a row: no terminator between the
    copies, so common prefixes run to (copies - 1) x the passage.  Returns (symbols uint32, n_strings = 1)."""
    passage = rng.integers(65, 91, size=int(passage_symbols), dtype=np.uint32)
    out = np.concatenate([np.tile(passage, int(copies)), np.array([TERMINATOR_START], dtype=np.uint32)])
    return out, 1


# ---- prose-like text from a character model ---------------------------------------------------------
# BASELINE config 5 names enwik8, which is not available offline.  The Zipf stand-in above has the word statistics of
# natural language but uniform letters; this one has the letters too: an order-3 character model (the 8 likeliest
# successors of every 3-character context) trained on the prose that ships with the container image
# (tools/train_prose_model.py -> east/data/prose_order3.npz, committed, so that the generator gives the same bytes on
# any machine).  Text comes out of many short chains run side by side (numpy), each started at a word start.
PROSE_ALPHABET = "abcdefghijklmnopqrstuvwxyz \n0123456789.,;:'\"-()!?"
PROSE_ALPHABET_SIZE = len(PROSE_ALPHABET)
_prose_model = None


def prose_model():
    """The committed order-3 model as sampling tables over the contexts that occur (numbered 0 .. C-1): pick[c, u] = which
    of the context's 8 likeliest successors a random byte u selects (probabilities in 1/256), succ[c, k] that character,
    next_ctx[c, k] the context that follows (-1: never seen with a successor), the word-start contexts with their distribution."""
    global _prose_model
    if _prose_model is None:
        import os
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "prose_order3.npz"))
        A = PROSE_ALPHABET_SIZE
        contexts = z["contexts"].astype(np.int64)               # sorted ids (c1 * A + c2) * A + c3
        thr = np.minimum(np.floor(z["cum"].astype(np.float64) * 256.0), 255).astype(np.int64)
        pick = (thr[:, None, :] < np.arange(256)[None, :, None]).sum(axis=2).astype(np.uint8)    # [C, 256] inverse CDF: which of the 8
        succ = z["succ"].astype(np.uint8)                                                        # [C, 8]
        follow = (contexts[:, None] % (A * A)) * A + succ.astype(np.int64)                       # [C, 8] ids of the following contexts
        pos = np.searchsorted(contexts, follow).clip(0, contexts.size - 1)
        next_ctx = np.where(contexts[pos] == follow, pos, -1).astype(np.int32)
        starts = np.searchsorted(contexts, z["starts"].astype(np.int64))
        _prose_model = {"pick": pick.reshape(-1), "succ": succ.reshape(-1), "next_ctx": next_ctx.reshape(-1), "contexts": contexts,
                        "starts": starts, "start_cdf": z["start_cdf"],
                        "chars": np.frombuffer(PROSE_ALPHABET.encode("ascii"), dtype=np.uint8)}
    return _prose_model


def prose_like_texts(rng, n_docs, doc_bytes, chain=256):
    """n_docs raw texts (bytes, ASCII: lower-case prose with punctuation and line breaks; 8 % of the words capitalised) of
    doc_bytes each, from the character model.  All documents are generated together: chains of `chain` characters side
    by side, two table look-ups per chain and character."""
    m = prose_model()
    A = PROSE_ALPHABET_SIZE
    per_doc = -(-doc_bytes // chain)
    n_chains = n_docs * per_doc

    def fresh(k):
        return m["starts"][np.searchsorted(m["start_cdf"], rng.random(k), side="left").clip(0, m["starts"].size - 1)]

    ctx = fresh(n_chains).astype(np.int64)
    out = np.empty((chain, n_chains), dtype=np.uint8)          # (a row per step: contiguous writes; transposed at the end)
    ids = m["contexts"][ctx]                                   # a start context is (blank, c1, c2): the chain opens with c1 c2
    out[0] = (ids // A) % A
    out[1] = ids % A
    for t in range(2, chain):
        k = ctx * 8 + m["pick"][ctx * 256 + rng.integers(0, 256, size=n_chains)]
        out[t] = m["succ"][k]
        ctx = m["next_ctx"][k].astype(np.int64)
        dead = ctx < 0
        if dead.any():                                         # (a context seen only at the very end of the training text)
            ctx[dead] = fresh(int(dead.sum()))
    text = np.ascontiguousarray(m["chars"][out].T)             # [n_chains, chain] bytes
    text[:, -1] = 32                                           # chains are joined by a blank
    # capitals: the first letter of 8 % of the words
    flat = text.reshape(-1)
    word_start = np.ones(flat.size, dtype=bool)
    word_start[1:] = (flat[:-1] == 32) | (flat[:-1] == 10)
    cap = np.flatnonzero(word_start & (flat >= 97) & (flat <= 122))
    flat[cap[rng.random(cap.size) < 0.08]] -= 32
    docs = text.reshape(n_docs, per_doc * chain)[:, :doc_bytes]
    return [bytes(row) for row in docs]
