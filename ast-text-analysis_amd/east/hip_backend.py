# -*- coding: utf-8 -*-
"""ctypes binding of the C ABI in include/east_hip.h (libeast_hip.so).

Thin by design: numpy arrays in, numpy arrays out, every error code turned
into HipBackendError.  There is no CPU fallback -- if the library or a HIP
device is missing, every compute call raises.
"""
import atexit
import ctypes
import os

import numpy as np

from east import exceptions

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EAST_HIP_LIBRARY", os.path.join(_HERE, "_lib", "libeast_hip.so"))

_c_i64p = ctypes.POINTER(ctypes.c_int64)
_c_i32p = ctypes.POINTER(ctypes.c_int32)
_c_u32p = ctypes.POINTER(ctypes.c_uint32)
_c_u64p = ctypes.POINTER(ctypes.c_uint64)
_c_dblp = ctypes.POINTER(ctypes.c_double)

# name -> (restype, argtypes); mirrors include/east_hip.h one to one
SIGNATURES = {
    "east_hip_version": (ctypes.c_char_p, []),
    "east_hip_last_error": (ctypes.c_char_p, []),
    "east_hip_device_count": (ctypes.c_int, []),
    "east_hip_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p)]),
    "east_hip_reset": (ctypes.c_int, [ctypes.c_void_p]),
    "east_hip_destroy": (None, [ctypes.c_void_p]),
    "east_hip_build": (ctypes.c_int, [ctypes.c_void_p, _c_u32p, ctypes.c_int64, _c_i64p, _c_i32p, ctypes.c_int32]),
    "east_hip_build_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, _c_i64p, _c_i32p,
                                             ctypes.c_int32]),
    "east_hip_build_texts": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, _c_i64p, ctypes.c_int32,
                                            ctypes.POINTER(ctypes.c_uint8), _c_u32p, _c_u32p, _c_u32p, _c_u32p,
                                            _c_u32p, ctypes.c_int32]),
    "east_hip_build_texts_v": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p), _c_i64p, ctypes.c_int32,
                                              ctypes.POINTER(ctypes.c_uint8), _c_u32p, _c_u32p, _c_u32p, _c_u32p,
                                              _c_u32p, ctypes.c_int32]),
    "east_hip_get_prepared": (ctypes.c_int, [ctypes.c_void_p, _c_i64p, _c_i64p, _c_i32p, _c_u32p]),
    "east_hip_prepared_encoding": (ctypes.c_int, [ctypes.c_void_p]),
    "east_hip_set_symbol_encoding": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "east_hip_last_prep_ms": (ctypes.c_double, [ctypes.c_void_p]),
    "east_hip_get_tables": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32] + [_c_i32p] * 6),
    "east_hip_score_table": (ctypes.c_int, [ctypes.c_void_p, _c_u32p, _c_i64p, ctypes.c_int32, ctypes.c_int,
                                            _c_dblp, _c_dblp]),
    "east_hip_score_table_grouped": (ctypes.c_int, [ctypes.c_void_p, _c_u32p, _c_i64p, ctypes.c_int32, _c_i64p,
                                                    ctypes.c_int32, ctypes.c_int, _c_dblp]),
    "east_hip_get_lcp_intervals": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, _c_i32p]),
    "east_hip_set_keyphrases": (ctypes.c_int, [ctypes.c_void_p, _c_u32p, _c_i64p, ctypes.c_int32]),
    "east_hip_score_resident": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "east_hip_score_probes": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_i64p]),
    "east_hip_score_resident_async": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "east_hip_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "east_hip_stream": (ctypes.c_void_p, [ctypes.c_void_p]),
    "east_hip_build_info": (ctypes.c_int, [ctypes.c_void_p, _c_i64p, ctypes.c_int32]),
    "east_hip_last_build_ms": (ctypes.c_double, [ctypes.c_void_p]),
    "east_hip_last_score_ms": (ctypes.c_double, [ctypes.c_void_p]),
    "east_hip_debug_radix_sort_u64": (ctypes.c_int, [ctypes.c_int, _c_u64p, _c_u32p, ctypes.c_int64, ctypes.c_int]),
    "east_hip_debug_radix_sort_u32": (ctypes.c_int, [ctypes.c_int, _c_u32p, _c_u32p, ctypes.c_int64, ctypes.c_int]),
    "east_hip_debug_exclusive_scan": (ctypes.c_int, [ctypes.c_int, _c_u32p, _c_u32p, ctypes.c_int64]),
    "east_hip_debug_suffix_array": (ctypes.c_int, [ctypes.c_int, _c_u32p, ctypes.c_int64, ctypes.c_uint32, _c_i32p,
                                                   _c_i32p]),
    "east_hip_plan_arena_bytes": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int32]),
    "east_hip_plan_arena_bytes_lean": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int32]),
    "east_hip_debug_set_rank_bucket_bytes": (ctypes.c_int, [ctypes.c_int64]),
    "east_hip_debug_set_window_sort": (ctypes.c_int, [ctypes.c_int]),
    "east_hip_debug_set_lds_rounds": (ctypes.c_int, [ctypes.c_int]),
    "east_hip_debug_set_persist": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "east_hip_debug_set_segmented_sort": (ctypes.c_int, [ctypes.c_int]),
    "east_hip_debug_set_speculation": (ctypes.c_int, [ctypes.c_int]),
    "east_hip_debug_set_score_scratch": (ctypes.c_int, [ctypes.c_int64]),
    "east_hip_debug_set_score_path": (ctypes.c_int, [ctypes.c_int]),
    "east_hip_debug_set_score_grid": (ctypes.c_int, [ctypes.c_int64]),
    "east_hip_debug_set_text_stream": (ctypes.c_int, [ctypes.c_int64]),
    "east_hip_debug_set_text_ring": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64]),
    "east_hip_debug_alphabetic_code": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int32,
                                                      ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int32)]),
    "east_hip_debug_narrow_symbols": (ctypes.c_int, [_c_u32p, ctypes.c_int64, ctypes.POINTER(ctypes.c_uint16), ctypes.c_int]),
    "east_hip_debug_narrow_symbols8": (ctypes.c_int, [_c_u32p, ctypes.c_int64, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int]),
    "east_hip_group_create": (ctypes.c_int, [_c_i32p, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]),
    "east_hip_group_destroy": (None, [ctypes.c_void_p]),
    "east_hip_group_build": (ctypes.c_int, [ctypes.c_void_p, _c_u32p, ctypes.c_int64, _c_i64p, _c_i32p, ctypes.c_int32,
                                            ctypes.c_int32]),
    "east_hip_group_build_texts_v": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p), _c_i64p, ctypes.c_int32,
                                                    ctypes.POINTER(ctypes.c_uint8), _c_u32p, _c_u32p, _c_u32p, _c_u32p,
                                                    _c_u32p, ctypes.c_int32]),
    "east_hip_group_shards": (ctypes.c_int, [ctypes.c_void_p, _c_i32p]),
    "east_hip_group_handle": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int32]),
    "east_hip_score_table_multi": (ctypes.c_int, [ctypes.c_void_p, _c_u32p, _c_i64p, ctypes.c_int32, ctypes.c_int, _c_dblp]),
    "east_hip_group_info": (ctypes.c_int, [ctypes.c_void_p, _c_dblp, ctypes.c_int32]),
    "east_hip_debug_shard_documents": (ctypes.c_int, [_c_i64p, ctypes.c_int32, ctypes.c_int32, _c_i32p]),
    "east_hip_format_table_xml": (ctypes.c_int64, [_c_dblp, ctypes.c_int32, ctypes.c_int32, _c_i32p, _c_i32p,
                                                   ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p),
                                                   ctypes.c_char_p, ctypes.c_int64]),
    "east_hip_format_table_csv": (ctypes.c_int64, [_c_dblp, ctypes.c_int32, ctypes.c_int32, _c_i32p, _c_i32p,
                                                   ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p),
                                                   ctypes.c_char_p, ctypes.c_int64]),
    "east_hip_profile_enable": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "east_hip_profile_only": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p]),
    "east_hip_profile_report": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64]),
}

BUILD_INFO_FIELDS = ("n_total", "n_docs", "n_strings", "sigma_text", "bits_level0", "dc3_levels", "arena_bytes",
                     "arena_high_water", "radix_passes", "radix_elements", "radix_element_bytes",
                     "radix_passes_u32", "radix_elements_u32", "radix_passes_u64", "radix_elements_u64",
                     "dc3_levels_resolved", "merge_elements", "refine_rounds", "window_sorted", "lds_sorted",
                     "fused_finish", "first_kept", "first_n", "ht_keys", "seg_sort", "narrow_upload", "persist_rounds")

_lib = None


def load():
    """Load libeast_hip.so (built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise exceptions.HipBackendError(
                reason="%s not found -- build it with `make -C ast-text-analysis_amd/csrc` "
                       "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback" % LIB_PATH)
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise exceptions.HipBackendError(reason="cannot load %s: %s" % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        msg = load().east_hip_last_error()
        raise exceptions.HipBackendError(reason="%s (code %d)" % (msg.decode("utf-8", "replace") if msg else "?", rc))


def _ptr(a, t):
    return a.ctypes.data_as(t)


def device_count():
    c = load().east_hip_device_count()
    return c if c > 0 else 0


def default_device():
    for var in ("EAST_HIP_DEVICE", "LOCAL_RANK"):
        if os.environ.get(var, "") != "":
            return int(os.environ[var])
    return 0


_unicode_tables = None


def unicode_tables():
    """The interpreter's own Unicode data for east_hip_build_texts: per code point below U+0A00 the
    class (bit 0: matches [\\w'] under re.U, i.e. str.isalnum() or '_' or "'"; bit 1: str.isdigit())
    and the 1:1 upper-case mapping; a bitmap of the word characters from U+0A00 up."""
    global _unicode_tables
    if _unicode_tables is None:
        lim = 0x0A00
        cls = np.zeros(lim, dtype=np.uint8)
        upper = np.arange(lim, dtype=np.uint32)
        for cp in range(lim):
            ch = chr(cp)
            cls[cp] = (1 if (ch.isalnum() or ch in "_'") else 0) | (2 if ch.isdigit() else 0)
            up = ch.upper()
            if len(up) == 1:
                upper[cp] = ord(up)
        n_hi = 0x110000 - lim
        wbits = np.zeros(n_hi, dtype=np.uint8)
        dbits = np.zeros(n_hi, dtype=np.uint8)
        hi_from, hi_to = [], []
        for cp in range(lim, 0x110000):
            ch = chr(cp)
            if ch.isalnum():
                wbits[cp - lim] = 1
                if ch.isdigit():
                    dbits[cp - lim] = 1
            up = ch.upper()
            if len(up) == 1 and up != ch:
                hi_from.append(cp)
                hi_to.append(ord(up))
        word_hi = np.packbits(wbits, bitorder="little").view(np.uint32).copy()
        digit_hi = np.packbits(dbits, bitorder="little").view(np.uint32).copy()
        _unicode_tables = (cls, upper, word_hi, digit_hi, np.array(hi_from, dtype=np.uint32),
                           np.array(hi_to, dtype=np.uint32))
    return _unicode_tables


JOIN_FREE_MAX_TEXTS = 512               # build_texts: up to this many texts of at least ...
JOIN_FREE_MIN_BYTES = 1 << 20           # ... this many bytes in all go to the device one by one, unjoined,
JOIN_FREE_RING_BYTES = 8 << 20          # ... and any number of texts from this size on (the streamed preparation: the
#                                         library copies them into a pinned ring with a few threads; joining 256 MB in
#                                         Python takes 50 ms, four times what the device needs for the whole index)
JOIN_FREE_RING_MIN_MEAN = 4096          # ... while a text averages this many bytes
JOIN_FREE_RING_MAX_TEXTS = 1 << 20
POOL_HANDLES = 2                        # recycled handles kept per device ...
POOL_MAX_ARENA_BYTES = 256 << 20        # ... if their device arena is at most this large
_handle_pool = {}


def _drain_pool():
    lib = _lib
    for handles in _handle_pool.values():
        while handles:
            if lib is not None:
                lib.east_hip_destroy(handles.pop())
            else:
                handles.pop()


atexit.register(_drain_pool)


class HipIndex(object):
    """One device-resident batch of annotated suffix arrays (an AST shard)."""

    def __init__(self, device=None, reserve_symbols=0):
        self._lib = load()
        self._h = ctypes.c_void_p()
        self.device = default_device() if device is None else int(device)
        pooled = _handle_pool.get(self.device)
        if pooled and not reserve_symbols:
            self._h = pooled.pop()               # a recycled handle (reset: behaves like a new one)
        else:
            _check(self._lib.east_hip_create(self.device, int(reserve_symbols), ctypes.byref(self._h)))
        self.n_docs = 0
        self.doc_offsets = None
        self._host_symbols = None

    @classmethod
    def borrowed(cls, handle, device):
        """A view of a handle somebody else owns (a shard of a HipGroup): close() leaves it alone."""
        self = cls.__new__(cls)
        self._lib = load()
        self._h = ctypes.c_void_p(handle)
        self._borrowed = True
        self.device = int(device)
        self.n_docs = 0
        self.doc_offsets = None
        self._host_symbols = None
        return self

    def close(self):
        """Release the handle.  Handles of small indexes go back to a per-device pool instead of being
        destroyed: stream / event / memory set-up costs more than building a small collection."""
        h = getattr(self, "_h", None)
        if not h:
            return
        self._h = None
        if getattr(self, "_borrowed", False):
            return
        try:
            pool = _handle_pool.setdefault(self.device, [])
            buf = np.zeros(8, dtype=np.int64)
            keep = (len(pool) < POOL_HANDLES and self._lib.east_hip_build_info(h, _ptr(buf, _c_i64p), buf.size) > 6
                    and buf[6] <= POOL_MAX_ARENA_BYTES and self._lib.east_hip_reset(h) == 0)
        except Exception:                        # interpreter shutdown: module globals may be gone
            keep = False
        if keep:
            pool.append(h)
        else:
            self._lib.east_hip_destroy(h)

    __del__ = close

    # -- build ---------------------------------------------------------------
    def build(self, symbols, doc_offsets, n_strings):
        """symbols: the reference encoding (terminator i of a document = 0x0A00+i, text below), or -- told
        by the tag bit of the last symbol -- the tagged one (asts/utils.py: strings_to_symbols)."""
        symbols = np.ascontiguousarray(symbols, dtype=np.uint32)
        doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.int64)
        n_strings = np.ascontiguousarray(n_strings, dtype=np.int32)
        _check(self._lib.east_hip_set_symbol_encoding(self._h, 1 if symbols.size and int(symbols[-1]) >> 31 else 0))
        _check(self._lib.east_hip_build(self._h, _ptr(symbols, _c_u32p), symbols.size, _ptr(doc_offsets, _c_i64p),
                                        _ptr(n_strings, _c_i32p), n_strings.size))
        self.n_docs = int(n_strings.size)
        self.doc_offsets = doc_offsets.copy()
        self._host_symbols = symbols

    def build_device(self, d_symbols_ptr, n_total, doc_offsets, n_strings, tagged=False):
        """d_symbols_ptr: integer address of a uint32 device buffer (e.g. tensor.data_ptr())."""
        doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.int64)
        n_strings = np.ascontiguousarray(n_strings, dtype=np.int32)
        _check(self._lib.east_hip_set_symbol_encoding(self._h, 1 if tagged else 0))
        _check(self._lib.east_hip_build_device(self._h, ctypes.c_void_p(int(d_symbols_ptr)), int(n_total),
                                               _ptr(doc_offsets, _c_i64p), _ptr(n_strings, _c_i32p), n_strings.size))
        self.n_docs = int(n_strings.size)
        self.doc_offsets = doc_offsets.copy()
        self._host_symbols = None

    def build_texts(self, texts):
        """Text preparation + build on the device.  texts: list of bytes (UTF-8, decoded with
        errors='replace' semantics) or str."""
        raw = [t if isinstance(t, bytes) else t.encode("utf-8", errors="surrogatepass") for t in texts]
        cls, upper, word_hi, digit_hi, hi_from, hi_to = unicode_tables()
        tables = (_ptr(cls, ctypes.POINTER(ctypes.c_uint8)), _ptr(upper, _c_u32p), _ptr(word_hi, _c_u32p),
                  _ptr(digit_hi, _c_u32p), _ptr(hi_from, _c_u32p), _ptr(hi_to, _c_u32p), hi_from.size)
        total = sum(len(t) for t in raw)
        # (... through the pinned ring only while a text averages JOIN_FREE_RING_MIN_MEAN bytes: half a million one-line
        # texts cost more as a ctypes pointer array than the one b"".join they would save)
        if (len(raw) <= JOIN_FREE_MAX_TEXTS and total >= JOIN_FREE_MIN_BYTES) or \
                (len(raw) <= JOIN_FREE_RING_MAX_TEXTS and total >= JOIN_FREE_RING_BYTES and
                 total >= JOIN_FREE_RING_MIN_MEAN * len(raw)):
            # a few large texts: uploaded one by one straight out of their bytes objects (joining 64 MiB costs
            # more host time than the device needs for the whole build)
            ptrs = (ctypes.c_char_p * len(raw))(*raw)
            lengths = np.array([len(t) for t in raw], dtype=np.int64)
            _check(self._lib.east_hip_build_texts_v(self._h, ptrs, _ptr(lengths, _c_i64p), len(raw), *tables))
        else:
            blob = b"\xff".join(raw + [b""])    # every text followed by one 0xFF, in a single copy
            offsets = np.zeros(len(raw) + 1, dtype=np.int64)
            np.cumsum([len(t) + 1 for t in raw], out=offsets[1:])
            _check(self._lib.east_hip_build_texts(self._h, blob, len(blob), _ptr(offsets, _c_i64p), len(raw), *tables))
        self.n_docs = len(raw)
        self._host_symbols = None
        doc_offsets = np.zeros(self.n_docs + 1, dtype=np.int64)
        n_total = ctypes.c_int64(0)
        _check(self._lib.east_hip_get_prepared(self._h, ctypes.byref(n_total), _ptr(doc_offsets, _c_i64p), None, None))
        self.doc_offsets = doc_offsets

    def prepared(self):
        """(symbols uint32, doc_offsets int64, n_strings int32) of the last build_texts.  The symbols are in
        the reference encoding (terminator i = 0x0A00+i) unless kept tokens hold characters at or above
        U+0A00: then in the tagged one (asts/utils.py: is_tagged / reference_code_points)."""
        n_total = ctypes.c_int64(0)
        _check(self._lib.east_hip_get_prepared(self._h, ctypes.byref(n_total), None, None, None))
        doc_offsets = np.zeros(self.n_docs + 1, dtype=np.int64)
        n_strings = np.zeros(self.n_docs, dtype=np.int32)
        symbols = np.zeros(n_total.value, dtype=np.uint32)
        _check(self._lib.east_hip_get_prepared(self._h, ctypes.byref(n_total), _ptr(doc_offsets, _c_i64p),
                                               _ptr(n_strings, _c_i32p), _ptr(symbols, _c_u32p)))
        return symbols, doc_offsets, n_strings

    def symbols(self):
        """The symbols of the index as the host last saw them: kept by build(), fetched from the device
        after build_texts()."""
        if self._host_symbols is None:
            self._host_symbols = self.prepared()[0]
        return self._host_symbols

    @property
    def last_prep_ms(self):
        return float(self._lib.east_hip_last_prep_ms(self._h))

    def tables(self, doc=0, names=("suftab", "lcptab", "anntab", "childtab_up", "childtab_down",
                                   "childtab_next_l_index")):
        order = ("suftab", "lcptab", "anntab", "childtab_up", "childtab_down", "childtab_next_l_index")
        if self.doc_offsets is None or not 0 <= doc < self.n_docs:
            raise exceptions.HipBackendError(reason="no index has been built on this handle, or no such document")
        nd = int(self.doc_offsets[doc + 1] - self.doc_offsets[doc])
        bufs = {k: np.empty(nd, dtype=np.int32) for k in names}
        args = [_ptr(bufs[k], _c_i32p) if k in bufs else None for k in order]
        _check(self._lib.east_hip_get_tables(self._h, int(doc), *args))
        return {k: v.astype(np.int64) for k, v in bufs.items()}    # the reference's np.int dtype

    # -- score ---------------------------------------------------------------
    def score_table(self, q_symbols, q_offsets, normalized=True, want_suffix=False):
        q_symbols = np.ascontiguousarray(q_symbols, dtype=np.uint32)
        q_offsets = np.ascontiguousarray(q_offsets, dtype=np.int64)
        K = q_offsets.size - 1
        out = np.empty((K, self.n_docs), dtype=np.float64)
        suf = np.empty((self.n_docs, int(q_offsets[-1])), dtype=np.float64) if want_suffix else None
        _check(self._lib.east_hip_score_table(self._h, _ptr(q_symbols, _c_u32p), _ptr(q_offsets, _c_i64p), K,
                                              int(bool(normalized)), _ptr(out, _c_dblp),
                                              _ptr(suf, _c_dblp) if want_suffix else None))
        return (out, suf) if want_suffix else out

    def score_table_grouped(self, q_symbols, q_offsets, group_offsets, normalized=True):
        """Scores of groups of queries: out[g, d] = max over the queries of group g (synonym variants)."""
        q_symbols = np.ascontiguousarray(q_symbols, dtype=np.uint32)
        q_offsets = np.ascontiguousarray(q_offsets, dtype=np.int64)
        group_offsets = np.ascontiguousarray(group_offsets, dtype=np.int64)
        G = group_offsets.size - 1
        out = np.empty((G, self.n_docs), dtype=np.float64)
        _check(self._lib.east_hip_score_table_grouped(self._h, _ptr(q_symbols, _c_u32p), _ptr(q_offsets, _c_i64p),
                                                      q_offsets.size - 1, _ptr(group_offsets, _c_i64p), G,
                                                      int(bool(normalized)), _ptr(out, _c_dblp)))
        return out

    def lcp_interval_lefts(self, doc=0):
        """left[k] = left boundary of the lcp-interval whose first l-index is rank k, -1 for other ranks."""
        if self.doc_offsets is None or not 0 <= doc < self.n_docs:
            raise exceptions.HipBackendError(reason="no index has been built on this handle, or no such document")
        left = np.empty(int(self.doc_offsets[doc + 1] - self.doc_offsets[doc]), dtype=np.int32)
        _check(self._lib.east_hip_get_lcp_intervals(self._h, int(doc), _ptr(left, _c_i32p)))
        return left.astype(np.int64)

    def set_keyphrases(self, q_symbols, q_offsets):
        q_symbols = np.ascontiguousarray(q_symbols, dtype=np.uint32)
        q_offsets = np.ascontiguousarray(q_offsets, dtype=np.int64)
        _check(self._lib.east_hip_set_keyphrases(self._h, _ptr(q_symbols, _c_u32p), _ptr(q_offsets, _c_i64p),
                                                 q_offsets.size - 1))

    def score_resident(self, normalized=True, d_out_ptr=None):
        _check(self._lib.east_hip_score_resident(self._h, int(bool(normalized)),
                                                 ctypes.c_void_p(int(d_out_ptr)) if d_out_ptr else None))

    def score_probes(self, normalized=True):
        """Table reads + binary-search probes of one pass of the score walk over the resident keyphrases."""
        c = ctypes.c_int64(0)
        _check(self._lib.east_hip_score_probes(self._h, int(bool(normalized)), ctypes.byref(c)))
        return int(c.value)

    def synchronize(self):
        _check(self._lib.east_hip_synchronize(self._h))

    @property
    def stream(self):
        return self._lib.east_hip_stream(self._h)

    def info(self):
        buf = np.zeros(len(BUILD_INFO_FIELDS), dtype=np.int64)
        self._lib.east_hip_build_info(self._h, _ptr(buf, _c_i64p), buf.size)
        return dict(zip(BUILD_INFO_FIELDS, (int(x) for x in buf)))

    def profile_enable(self, on=True):
        _check(self._lib.east_hip_profile_enable(self._h, int(bool(on))))

    def profile_only(self, kernel=None):
        """Bracket only the launches of `kernel` (None: every kernel) while profiling is enabled."""
        _check(self._lib.east_hip_profile_only(self._h, kernel.encode("ascii") if kernel else None))

    def profile_report(self):
        """{kernel name: (launches, total_ms)} accumulated since profile_enable()."""
        buf = ctypes.create_string_buffer(1 << 16)
        self._lib.east_hip_profile_report(self._h, buf, len(buf))
        out = {}
        for line in buf.value.decode().splitlines():
            name, count, ms = line.split("\t")
            out[name.strip("()")] = (int(count), float(ms))
        return out

    @property
    def last_build_ms(self):
        return float(self._lib.east_hip_last_build_ms(self._h))

    @property
    def last_score_ms(self):
        return float(self._lib.east_hip_last_score_ms(self._h))


class HipGroup(object):
    """One collection over several devices in this process: a shard of documents -- one HipIndex -- per device, driven
    by the library's own host threads, the K x D_local score blocks assembled by one all-gather (RCCL between distinct
    devices; include/east_hip.h, "Several devices in one process").  `devices`: a count (devices 0 .. N-1) or a list of
    ordinals; an ordinal may appear several times (logical shards on one device)."""

    def __init__(self, devices):
        self._lib = load()
        self.devices = list(range(devices)) if isinstance(devices, int) else [int(d) for d in devices]
        if not self.devices:
            raise exceptions.HipBackendError(reason="a device group needs at least one device")
        self._g = ctypes.c_void_p()
        dev = np.array(self.devices, dtype=np.int32)
        _check(self._lib.east_hip_group_create(_ptr(dev, _c_i32p), dev.size, ctypes.byref(self._g)))
        self.n_docs = 0
        self.first_doc = None
        self.shards = []

    def close(self):
        g = getattr(self, "_g", None)
        if g:
            self._g = None
            for shard in self.shards:
                shard.close()
            self._lib.east_hip_group_destroy(g)

    __del__ = close

    def _after_build(self, n_docs):
        self.n_docs = int(n_docs)
        first = np.zeros(len(self.devices) + 1, dtype=np.int32)
        self._lib.east_hip_group_shards(self._g, _ptr(first, _c_i32p))
        self.first_doc = first
        self.shards = []
        for s, device in enumerate(self.devices):
            view = HipIndex.borrowed(self._lib.east_hip_group_handle(self._g, s), device)
            view.n_docs = int(first[s + 1] - first[s])
            if view.n_docs:
                offsets = np.zeros(view.n_docs + 1, dtype=np.int64)
                n_total = ctypes.c_int64(0)
                if self._from_texts:
                    _check(self._lib.east_hip_get_prepared(view._h, ctypes.byref(n_total), _ptr(offsets, _c_i64p), None, None))
                else:
                    b, e = int(first[s]), int(first[s + 1])
                    offsets = (self._doc_offsets[b:e + 1] - self._doc_offsets[b]).astype(np.int64)
                    view._host_symbols = self._symbols[int(self._doc_offsets[b]):int(self._doc_offsets[e])]
                view.doc_offsets = offsets
            self.shards.append(view)

    def build(self, symbols, doc_offsets, n_strings):
        symbols = np.ascontiguousarray(symbols, dtype=np.uint32)
        doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.int64)
        n_strings = np.ascontiguousarray(n_strings, dtype=np.int32)
        tagged = 1 if symbols.size and int(symbols[-1]) >> 31 else 0
        _check(self._lib.east_hip_group_build(self._g, _ptr(symbols, _c_u32p), symbols.size, _ptr(doc_offsets, _c_i64p),
                                              _ptr(n_strings, _c_i32p), n_strings.size, tagged))
        self._from_texts, self._symbols, self._doc_offsets = False, symbols, doc_offsets
        self._after_build(n_strings.size)

    def build_texts(self, texts):
        raw = [t if isinstance(t, bytes) else t.encode("utf-8", errors="surrogatepass") for t in texts]
        cls, upper, word_hi, digit_hi, hi_from, hi_to = unicode_tables()
        ptrs = (ctypes.c_char_p * len(raw))(*raw)
        lengths = np.array([len(t) for t in raw], dtype=np.int64)
        _check(self._lib.east_hip_group_build_texts_v(
            self._g, ptrs, _ptr(lengths, _c_i64p), len(raw), _ptr(cls, ctypes.POINTER(ctypes.c_uint8)), _ptr(upper, _c_u32p),
            _ptr(word_hi, _c_u32p), _ptr(digit_hi, _c_u32p), _ptr(hi_from, _c_u32p), _ptr(hi_to, _c_u32p), hi_from.size))
        self._from_texts = True
        self._after_build(len(raw))

    def locate(self, doc):
        """(shard view, document number inside the shard) of document `doc` of the collection."""
        s = int(np.searchsorted(self.first_doc, doc, side="right")) - 1
        return self.shards[s], int(doc - self.first_doc[s])

    def score_table(self, q_symbols, q_offsets, normalized=True):
        q_symbols = np.ascontiguousarray(q_symbols, dtype=np.uint32)
        q_offsets = np.ascontiguousarray(q_offsets, dtype=np.int64)
        K = q_offsets.size - 1
        out = np.empty((K, self.n_docs), dtype=np.float64)
        _check(self._lib.east_hip_score_table_multi(self._g, _ptr(q_symbols, _c_u32p), _ptr(q_offsets, _c_i64p), K,
                                                    int(bool(normalized)), _ptr(out, _c_dblp)))
        return out

    def info(self):
        buf = np.zeros(5, dtype=np.float64)
        self._lib.east_hip_group_info(self._g, _ptr(buf, _c_dblp), buf.size)
        return {"build_ms": float(buf[0]), "score_ms": float(buf[1]), "gather_ms": float(buf[2]),
                "gather": {1: "rccl", 2: "copies"}.get(int(buf[3]), None), "shards": int(buf[4])}


def shard_documents(sizes, n_shards):
    """The library's sharding rule (host only): first_doc[n_shards + 1]."""
    sizes = np.ascontiguousarray(sizes, dtype=np.int64)
    first = np.zeros(n_shards + 1, dtype=np.int32)
    _check(load().east_hip_debug_shard_documents(_ptr(sizes, _c_i64p), sizes.size, n_shards, _ptr(first, _c_i32p)))
    return first


def format_table(scores, kp_order, text_order, kp_names, text_names, kind):
    """The keyphrase table as text through the library's host-side formatter (east_hip_format_table_xml / _csv): scores
    K x D float64, the output order of rows / columns as index arrays, names as str (CSV: already quoted)."""
    lib = load()
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    K, D = scores.shape
    kp_order = np.ascontiguousarray(kp_order, dtype=np.int32)
    text_order = np.ascontiguousarray(text_order, dtype=np.int32)
    kp_b = [s.encode("utf-8", "surrogatepass") for s in kp_names]
    text_b = [s.encode("utf-8", "surrogatepass") for s in text_names]
    if any(b"\0" in b for b in kp_b) or any(b"\0" in b for b in text_b):
        raise ValueError("NUL in a name")
    fn = lib.east_hip_format_table_xml if kind == "xml" else lib.east_hip_format_table_csv
    cap = 64 + sum(len(b) + 64 for b in kp_b) + (sum(len(b) + 64 for b in text_b) + 8) * max(K, 1) + 40 * K * D
    buf = ctypes.create_string_buffer(cap)
    n = fn(_ptr(scores, _c_dblp), K, D, _ptr(kp_order, _c_i32p), _ptr(text_order, _c_i32p),
           (ctypes.c_char_p * max(K, 1))(*kp_b), (ctypes.c_char_p * max(D, 1))(*text_b), buf, cap)
    if n < 0:
        raise exceptions.HipBackendError(reason="table formatter: %d" % n)
    return buf.raw[:n].decode("utf-8", "surrogatepass")


def pack_queries(queries, keep_spaces=False):
    """[unicode query] -> (q_symbols uint32, q_offsets int64); U+0020 is removed as score() does
    (easa.py:36) unless keep_spaces (synonym variants go to _score as they are, easa.py:34)."""
    from east.asts import utils as ast_utils
    parts = [ast_utils.query_to_symbols(q, keep_spaces) for q in queries]
    offsets = np.zeros(len(parts) + 1, dtype=np.int64)
    for i, p in enumerate(parts):
        offsets[i + 1] = offsets[i] + p.size
    symbols = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
    return symbols.astype(np.uint32), offsets
