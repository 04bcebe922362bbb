# -*- coding: utf-8 -*-
"""AST helpers (reference east/asts/utils.py:6-40)."""
import numpy as np

from east import consts
from east import exceptions


def index(array, key, start=0):
    """Linear scan without boundary check (asts/utils.py:6-11)."""
    i = start
    while array[i] != key:
        i += 1
    return i


def match_strings(str1, str2):
    """Largest i such that str1[:i] == str2[:i] (asts/utils.py:14-22)."""
    i = 0
    min_len = len(str1) if len(str1) < len(str2) else len(str2)
    while i < min_len and str1[i] == str2[i]:
        i += 1
    return i


def make_unique_endings(strings_collection):
    """String i gets the terminator chr(0x0A00 + i) appended (asts/utils.py:25-40)."""
    start = consts.String.UNICODE_SPECIAL_SYMBOLS_START
    return [s + chr(start + i) for i, s in enumerate(strings_collection)]


TERMINATOR_TAG = 0x80000000          # include/east_hip.h: EAST_HIP_TERMINATOR_TAG


def is_tagged(symbols):
    """Every document ends in a terminator, so the last symbol tells the encoding."""
    return bool(len(symbols)) and bool(int(symbols[-1]) & TERMINATOR_TAG)


def tag_terminators(symbols):
    """Reference encoding (terminator i = 0x0A00+i, text below) -> tagged encoding."""
    start = consts.String.UNICODE_SPECIAL_SYMBOLS_START
    symbols = np.asarray(symbols, dtype=np.uint32)
    if is_tagged(symbols):
        return symbols
    return np.where(symbols >= start, (symbols - np.uint32(start)) | np.uint32(TERMINATOR_TAG), symbols).astype(np.uint32)


def reference_code_points(symbols):
    """The code points of the reference's `self.string` for either encoding: terminator i is 0x0A00+i."""
    symbols = np.asarray(symbols, dtype=np.uint32)
    if not is_tagged(symbols):
        return symbols
    start = consts.String.UNICODE_SPECIAL_SYMBOLS_START
    return np.where(symbols >= TERMINATOR_TAG, (symbols & np.uint32(TERMINATOR_TAG - 1)) + np.uint32(start), symbols)


def strings_to_symbols(strings_collection, tagged=None):
    """The code points of "".join(make_unique_endings(strings)) as uint32 -- the
    input layout of east_hip_build (include/east_hip.h).  Terminators are written
    numerically, so collections beyond the reference's 1 111 552-string limit
    (0x0A00+i > 0x10FFFF) are fine.

    Text below U+0A00 gives the reference's own encoding (terminator i = 0x0A00+i).  Text at or above
    U+0A00 gives the TAGGED encoding (terminator i = TERMINATOR_TAG | i; `tagged=True` asks for it
    whatever the text holds), in which the terminators stay above every text symbol (include/east_hip.h:
    east_hip_set_symbol_encoding).  `tagged=False` insists on the reference encoding and raises
    SymbolOutOfDomainException for text it cannot hold."""
    start = consts.String.UNICODE_SPECIAL_SYMBOLS_START
    m = len(strings_collection)
    text = "".join(strings_collection)
    lens = np.fromiter((len(s) for s in strings_collection), dtype=np.int64, count=m)
    cps = np.frombuffer(text.encode("utf-32-le", errors="surrogatepass"), dtype="<u4")
    high = bool(cps.size) and int(cps.max()) >= start
    if high and tagged is False:
        raise exceptions.SymbolOutOfDomainException(code=int(cps[cps >= start][0]))
    if tagged is None:
        tagged = high
    out = np.empty(cps.size + m, dtype=np.uint32)
    term_pos = np.cumsum(lens) + np.arange(m)
    mask = np.ones(out.size, dtype=bool)
    mask[term_pos] = False
    out[mask] = cps
    out[term_pos] = np.arange(m, dtype=np.uint32) + np.uint32(start)
    if tagged:
        out[term_pos] = np.arange(m, dtype=np.uint32) | np.uint32(TERMINATOR_TAG)
    return out


def query_to_symbols(query, keep_spaces=False):
    """score() removes U+0020 only (easa.py:36)."""
    q = query if keep_spaces else query.replace(" ", "")
    return np.frombuffer(q.encode("utf-32-le", errors="surrogatepass"), dtype="<u4").astype(np.uint32)
