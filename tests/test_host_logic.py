"""CPU tier: host-side logic of the product package, the C-ABI surface, and the
"no silent fallback" rule.  No compute call needs a GPU here."""
import ctypes
import io
import json
import os
import re
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest

from conftest import ROOT, load_golden


def test_text_preparation_vectors():
    """tokenize / prepare_text / text_to_strings_collection / make_unique_endings against the
    vectors captured from the reference (incl. tests/test_utils.py:10-13)."""
    from east import utils
    from east.asts import utils as ast_utils
    g = load_golden("utils_vectors.json")
    assert utils.tokenize("Well, what a sunny day!") == ["Well", "what", "a", "sunny", "day"]
    for v in g["tokenize"]:
        assert utils.tokenize(v["text"]) == v["out"]
    for v in g["prepare_text"]:
        assert utils.prepare_text(v["text_utf8"].encode("utf-8")) == v["out"]
        assert utils.prepare_text(v["text_utf8"]) == v["out"]
    for v in g["text_to_strings_collection"]:
        assert utils.text_to_strings_collection(v["text_utf8"].encode("utf-8")) == v["out"]
    for v in g["match_strings"]:                         # tests/asts/test_utils.py:10-20
        assert ast_utils.match_strings(v["a"], v["b"]) == v["out"]
    for v in g["index"]:                                 # tests/asts/test_utils.py:22-30
        assert ast_utils.index(v["array"], v["key"]) == v["out"]
    for v in g["make_unique_endings"]:
        assert [[ord(c) for c in s] for s in ast_utils.make_unique_endings(v["strings"])] == v["out"]
        flat = [c for s in v["out"] for c in s]
        assert ast_utils.strings_to_symbols(v["strings"]).tolist() == flat


def test_prepare_text_is_code_point_to_code_point():
    from east import utils
    assert utils.prepare_text("straße") == "STRAßE"       # py2 unicode.upper() semantics of the reference
    assert utils.prepare_text(b"\xff\xfeabc") == "��ABC"


def test_strings_to_symbols_domain_and_large_collections():
    from east import exceptions
    from east.asts import utils as ast_utils
    # text at or above U+0A00: the tagged encoding (terminator i = TAG | i), unless the reference's is insisted on
    with pytest.raises(exceptions.SymbolOutOfDomainException):
        ast_utils.strings_to_symbols(["ok", "中"], tagged=False)
    sym = ast_utils.strings_to_symbols(["ok", "中"])
    assert sym.tolist() == [111, 107, 0x80000000, 0x4E2D, 0x80000001] and ast_utils.is_tagged(sym)
    assert ast_utils.reference_code_points(sym).tolist() == [111, 107, 0x0A00, 0x4E2D, 0x0A01]
    plain = ast_utils.strings_to_symbols(["ok", "no"])
    assert not ast_utils.is_tagged(plain) and plain.tolist() == [111, 107, 0x0A00, 110, 111, 0x0A01]
    assert ast_utils.tag_terminators(plain).tolist() == ast_utils.strings_to_symbols(["ok", "no"], tagged=True).tolist()
    assert ast_utils.tag_terminators(sym) is sym and ast_utils.reference_code_points(plain) is plain
    m = 1_200_000                                         # beyond the reference's 1 111 552-string limit
    sym = ast_utils.strings_to_symbols(["A"] * m)
    assert sym.size == 2 * m and int(sym[-1]) == 0x0A00 + m - 1 and int(sym[0]) == 65
    assert ast_utils.query_to_symbols("a b  c").tolist() == [97, 98, 99]


def test_synthetic_generator_matches_text_preparation():
    from east import synthetic, utils
    from east.asts import utils as ast_utils
    for n_bytes in [1, 2, 3, 5, 11, 12, 13, 50, 1000, 65536]:
        for seed in range(4):
            rng = np.random.default_rng(seed * 1000 + n_bytes)
            text, sym, m = synthetic.word_stream_document(rng, n_bytes)
            sc = utils.text_to_strings_collection(text)
            assert len(text) == n_bytes and m == len(sc)
            assert np.array_equal(ast_utils.strings_to_symbols(sc), sym)
    q, off = synthetic.keyphrases(np.random.default_rng(1), sym, 50)
    assert off[0] == 0 and off[-1] == q.size and (np.diff(off) > 0).all() and (q < 0x0A00).all()


def test_table_formats_match_reference_output():
    from east import formatting
    g = load_golden("sample_table.json")
    for mode in ("normalized", "denormalized"):
        assert formatting.table2xml(g[mode]) == g["xml_" + mode]
        assert formatting.table2csv(g[mode]) == g["csv_" + mode]
        assert formatting.format_table(g[mode], "xml") == g["xml_" + mode]
    h = load_golden("hse_config1.json")
    assert formatting.table2xml(h["normalized"]) == h["xml_normalized"]
    assert formatting.table2csv(h["normalized"]) == h["csv_normalized"]
    with pytest.raises(Exception):
        formatting.format_table(g["normalized"], "yaml")


def test_factory_contract_without_gpu_work():
    from east import consts, exceptions
    from east.asts import base
    with pytest.raises(exceptions.NoSuchASTAlgorithm) as e:
        base.AST.get_ast(["A"], "no_such_algorithm")
    assert "no_such_algorithm" in str(e.value)
    algs = {cls.__algorithm__ for cls in __import__("east").utils.itersubclasses(base.AST)}
    assert {"easa", "easa_hip", "ast_linear", "ast_naive"} <= algs       # auto-registration (east/__init__.py)
    assert consts.ASTAlgorithm.EASA == "easa" and consts.String.UNICODE_SPECIAL_SYMBOLS_START == 0x0A00
    assert consts.TraversalOrder.DEPTH_FIRST_PRE_ORDER == "depth-first|pre-order"


def _header_functions():
    with open(os.path.join(ROOT, "include", "east_hip.h")) as f:
        src = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(east_hip_\w+)\s*\(", src)))


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    from east import hip_backend
    if not os.path.exists(hip_backend.LIB_PATH):      # a fresh checkout: hipcc cross-compiles without a GPU
        import __graft_entry__
        __graft_entry__.build()
    lib = hip_backend.load()
    declared = _header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "libeast_hip.so does not export %s" % name
    assert sorted(hip_backend.SIGNATURES) == declared        # the binding covers the header one to one
    assert lib.east_hip_version().startswith(b"east-hip")
    assert lib.east_hip_plan_arena_bytes(1 << 20, 1) > (1 << 20) * 28
    assert lib.east_hip_plan_arena_bytes(0, 1) < 0


def test_no_silent_cpu_fallback():
    """Without a HIP device the product path fails loudly; it never routes to the oracle."""
    from east import exceptions, hip_backend
    from east.asts import base
    if hip_backend.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(exceptions.HipBackendError) as e:
        base.AST.get_ast(["XABXAC", "HI"])
    assert "no CPU fallback" in str(e.value)
    # empty collection is still the reference's exception, raised before any device work
    with pytest.raises(exceptions.EmptyStringsCollectionException):
        base.AST.get_ast([])
    pkg = os.path.join(ROOT, "ast-text-analysis_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip")):
                with open(os.path.join(dirpath, fn), encoding="utf-8") as f:
                    src = f.read()
                assert "easa_oracle" not in src and "import oracle" not in src and "from oracle" not in src, fn


def test_missing_library_is_a_loud_error(monkeypatch):
    from east import exceptions, hip_backend
    monkeypatch.setattr(hip_backend, "_lib", None)
    monkeypatch.setattr(hip_backend, "LIB_PATH", "/nonexistent/libeast_hip.so")
    with pytest.raises(exceptions.HipBackendError):
        hip_backend.load()


def test_cli_syntax_errors_return_1(tmp_path):
    from east import main
    for argv in ([], ["keyphrases"], ["foo", "bar"], ["keyphrases", "table"], ["keyphrases", "table", "x"]):
        buf = io.StringIO()
        with redirect_stdout(buf):
            assert main.main(argv) == 1
        assert "Invalid" in buf.getvalue()
    kp = tmp_path / "k.txt"
    kp.write_text("ABC\n")
    tx = tmp_path / "t.txt"
    tx.write_text("XABXAC\n")
    buf = io.StringIO()
    with redirect_stdout(buf):
        assert main.main(["-s", "cosine", "keyphrases", "table", str(kp), str(tx)]) == 1


def test_shard_documents():
    from east import parallel
    assert parallel.shard_documents([1] * 8, 2) == [(0, 4), (4, 8)]
    assert parallel.shard_documents([1] * 2048, 8) == [(256 * r, 256 * (r + 1)) for r in range(8)]
    for sizes, world in ([5, 1, 1, 1], 2), ([1], 4), ([], 2), ([3, 1, 4, 1, 5, 9, 2, 6], 3):
        shards = parallel.shard_documents(sizes, world)
        assert len(shards) == world and shards[0][0] == 0 and shards[-1][1] == len(sizes)
        assert all(shards[r][1] == shards[r + 1][0] for r in range(world - 1))


def _brute_force_lefts(lcp, ann):
    """left[k] = PSV(k) for the first l-indices (anntab > 0, k > 0), -1 elsewhere -- what
    east_hip_get_lcp_intervals returns, by plain scans."""
    left = np.full(len(lcp), -1, dtype=np.int64)
    for k in range(1, len(lcp)):
        if ann[k] > 0:
            p = k - 1
            while lcp[p] >= lcp[k]:
                p -= 1
            left[k] = p
    return left


def test_interval_traversals_from_closed_forms(oracle):
    """east/asts/intervals.py (the host half of the traversal API, easa.py:38-85) on the oracle's
    tables, against the visits recorded from the reference's traverse()."""
    from east.asts import intervals
    for case in load_golden("traversal_synonyms.json")["traversals"]:
        o = oracle.OracleEASA(case["strings"])
        left = _brute_force_lefts(o.lcptab, o.anntab)
        post = [[v[0], v[1], v[2], [c[:3] for c in v[3]]] for v in intervals.post_order(o.lcptab, o.anntab, left)]
        assert post == case["post_order"], case["strings"]
        last = None
        for last in intervals.post_order(o.lcptab, o.anntab, left):
            pass
        assert last == case["root_nested"]
        pre = [[v[0], v[1], v[2], ord(v[3]) if v[3] else -1]
               for v in intervals.pre_order(o.lcptab, o.anntab, left, o.childtab_down, o.suftab, o.symbols)]
        assert pre == case["pre_order"], case["strings"]


def test_synonym_variants():
    """The variants easa.py:27-33 scores: product of (synonyms + the word itself) per word."""
    from east import relevance

    class Syn(object):
        def get_synonyms(self):
            return {"QUICK": ["FAST", "RAPID"], "FOX": ["DOG"], "X": []}
    assert relevance.synonym_variants("QUICK FOX", Syn()) == ["FASTDOG", "FASTFOX", "RAPIDDOG", "RAPIDFOX", "QUICKDOG",
                                                             "QUICKFOX"]
    assert relevance.synonym_variants("X", Syn()) == ["X"]
    with pytest.raises(KeyError):
        relevance.synonym_variants("QUICK WOLF", Syn())
    with pytest.raises(ZeroDivisionError):
        relevance.synonym_variants(" ", Syn())


def test_order_preserving_variable_length_code():
    """csrc/ht_code.h (host side, no device): the code words of the first-level keys' variable-length code are ordered like
    the symbols, prefix-free, fill the code space exactly (Kraft sum 1), have 3 .. 12 bits, and -- where no length limit
    interferes -- cost what the optimal alphabetic tree costs (interval dynamic programme as the checker)."""
    from east import hip_backend
    lib = hip_backend.load()
    rng = np.random.default_rng(31)

    def code_of(weights):
        w = np.ascontiguousarray(weights, dtype=np.uint64)
        code = np.zeros(w.size, dtype=np.uint32)
        length = np.zeros(w.size, dtype=np.int32)
        rc = lib.east_hip_debug_alphabetic_code(w.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), w.size,
                                                code.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                                length.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        return rc, code, length

    def optimal_alphabetic_cost(w):
        n = len(w)
        pre = np.concatenate([[0], np.cumsum(w)])
        cost = [[0] * (n + 1) for _ in range(n + 1)]
        for span in range(2, n + 1):
            for i in range(0, n - span + 1):
                j = i + span
                cost[i][j] = min(cost[i][s] + cost[s][j] for s in range(i + 1, j)) + int(pre[j] - pre[i])
        return cost[0][n]

    cases = []
    for n in (8, 9, 27, 28, 57, 113, 255, 256):
        cases.append(rng.integers(1, 1000, size=n))
        cases.append(np.sort(rng.integers(1, 10**6, size=n))[::-1] ** 2)                 # steep
        zipf = (1e7 / np.arange(1, n + 1) ** 1.3).astype(np.int64) + 1
        cases.append(rng.permutation(zipf))
        cases.append(np.ones(n, dtype=np.int64))
    cases.append(np.array([10**9, 1, 1, 1, 1, 1, 1, 1, 1, 1]))                            # one symbol carries the text
    cases.append(np.array([1] * 200 + [10**12]))
    for w in cases:
        rc, code, length = code_of(w)
        assert rc == 0, (rc, len(w))
        assert length.min() >= 3 and length.max() <= 12
        left = [(int(c) << (12 - int(l))) for c, l in zip(code, length)]
        for i in range(len(w) - 1):
            assert left[i] + (1 << (12 - int(length[i]))) <= left[i + 1]               # ordered and prefix-free
        assert sum(1 << (12 - int(l)) for l in length) == 1 << 12                       # no hole in the code space
        if len(w) <= 57:
            want = optimal_alphabetic_cost(np.asarray(w, dtype=np.int64))
            got = int(sum(int(x) * int(l) for x, l in zip(w, length)))
            # (a case that runs into the 3 / 12 bit limits has its weights adjusted and may cost a little more: the steep
            # ones; flat weights never do)
            flat = int(np.max(w)) <= 1000 and length.min() > 3
            assert got >= want and (got == want or not flat), (len(w), got, want)
    assert code_of(np.array([1, 2, 3]))[0] != 0                                           # too few symbols: no code


def test_group_sharding_rule_matches_the_process_per_gpu_path():
    """The in-process device group (east_hip_group_*: csrc/multi.h) and the one-process-per-GPU path (east/parallel.py)
    cut a collection into the same contiguous blocks: balanced by size, possibly empty with fewer documents than shards."""
    from east import hip_backend, parallel
    rng = np.random.default_rng(5)
    for _ in range(200):
        n, g = int(rng.integers(0, 50)), int(rng.integers(1, 10))
        sizes = rng.integers(1, 5000, size=n)
        first = hip_backend.shard_documents(sizes, g).tolist()
        blocks = parallel.shard_documents(sizes, g)
        assert first == [b for b, _ in blocks] + [blocks[-1][1]]
        assert first[0] == 0 and first[-1] == n and all(a <= b for a, b in zip(first, first[1:]))


def test_prose_like_generator_is_deterministic_and_looks_like_prose():
    """synthetic.prose_like_texts (the order-3 character model committed under east/data/): the same bytes for the same
    seed -- the fixture tests/golden/prose_like_docs.json holds the documents of seed 20246 --, letter statistics of
    English text rather than uniform letters, and text the preparation chain turns into strings of three words."""
    import collections
    from east import synthetic, utils
    g = load_golden("prose_like_docs.json")
    docs = synthetic.prose_like_texts(np.random.default_rng(20246), 5, 6000)
    assert [d.decode("ascii") for d in docs[:4]] == [g["texts"]["doc%d" % i] for i in range(4)]
    big = b" ".join(synthetic.prose_like_texts(np.random.default_rng(1), 3, 100000))
    counts = collections.Counter(big.lower())
    letters = sorted((c for c in counts if 97 <= c <= 122), key=lambda c: -counts[c])
    assert set(bytes(letters[:6]).decode()) <= set("etaoinsrh") and counts[ord("e")] > 8 * counts[ord("z")]
    strings = utils.text_to_strings_collection(big)
    assert len(strings) > 8000 and all(s == s.upper() for s in strings[:100])


class _ArrayMeasure(object):
    """A batched measure that returns a given K x D array (keyphrases_table's fast path without a device)."""

    def __init__(self, scores):
        self.scores = scores

    def set_text_collection(self, texts, language=None):
        pass

    def relevance_table(self, prepared):
        return self.scores


def test_score_table_is_the_dict_of_dicts_and_the_graph_from_the_array_is_the_graph_of_the_loops(monkeypatch):
    """applications.ScoreTable (what keyphrases_table returns on the batched path: a mapping over the K x D array) equals
    the reference's dict of dicts (applications.py:46-52), and keyphrases_graph worked out on the array (matrix products)
    gives the nodes and edges of the reference's loops (applications.py:59-149), in their order -- repeated keyphrases,
    filtered nodes, self-pairs and zero-support sources included."""
    from east import applications
    rng = np.random.default_rng(5)
    for K, D in ((7, 5), (60, 17), (200, 33)):
        kps = ["kp%d" % i for i in range(K)]
        kps[3] = kps[1]                                       # a repeated keyphrase
        uniq = list(dict.fromkeys(kps))
        scores = rng.random((len(uniq), D)) * 0.5
        scores[0] = 0.0                                       # a keyphrase that occurs nowhere
        texts = {"t%d" % i: b"x" for i in range(D)}
        plain = {k: {t: float(scores[i, j]) for j, t in enumerate(texts)} for i, k in enumerate(uniq)}
        # the default for a table of this size is the reference's own type: a plain dict of dicts that can be changed
        # and serialised (applications.py:46-52) ...
        monkeypatch.undo()
        small = applications.keyphrases_table(kps, texts, _ArrayMeasure(scores))
        assert type(small) is dict and all(type(row) is dict for row in small.values()) and small == plain
        assert json.loads(json.dumps(small)) == plain
        small[uniq[0]]["t0"] = 2.0
        small["another"] = {}
        # ... and from ARRAY_TABLE_MIN_SCORES scores on, a mapping over the array
        monkeypatch.setattr(applications, "ARRAY_TABLE_MIN_SCORES", 0)
        table = applications.keyphrases_table(kps, texts, _ArrayMeasure(scores))
        assert isinstance(table, applications.ScoreTable)
        assert type(table.to_dict()) is dict and table.to_dict() == plain and json.loads(json.dumps(table.to_dict())) == plain
        assert table == plain and dict(table) == plain and sorted(table) == sorted(plain)
        assert list(table[uniq[2]].items()) == list(plain[uniq[2]].items())
        for support in (0, 1, 3):
            fast = applications.keyphrases_graph(kps, texts, 0.4, 0.25, support, _ArrayMeasure(scores))
            monkeypatch.setattr(applications, "keyphrases_table", lambda *a, **k: plain)
            slow = applications.keyphrases_graph(kps, texts, 0.4, 0.25, support, _ArrayMeasure(scores))
            monkeypatch.undo()
            monkeypatch.setattr(applications, "ARRAY_TABLE_MIN_SCORES", 0)
            assert fast == slow and len(fast["edges"]) > 0, (K, D, support)


def test_host_symbols_are_narrowed_to_16_bit_words_as_the_plain_loop_does():
    """east_hip_build sends host symbols of the reference encoding over the link as 16-bit words (east_hip.hip:
    upload_symbols_narrow): text below U+0A00 as it is, every terminator as 0xFFFF.  The host threads narrow with AVX2 and
    streaming stores where the CPU has them -- an alignment prologue, sixteen symbols a step, a tail --: every start
    alignment and length against numpy and against the library's plain loop (host only, no device)."""
    import ctypes
    from east import hip_backend
    lib = hip_backend.load()
    rng = np.random.default_rng(16)
    base = rng.integers(0, 0x0A00, size=5000).astype(np.uint32)
    special = np.array([0x09FF, 0x0A00, 0x0A01, 0xFFFF, 0x10000, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0], dtype=np.uint32)
    base[rng.integers(0, base.size, size=600)] = rng.choice(special, size=600)
    vector_ran = set()
    for start in range(0, 40):
        for n in (0, 1, 15, 16, 17, 31, 33, 100, 1000, 4096 + start):
            src = np.ascontiguousarray(base[start:start + n])
            want = np.where(src < 0x0A00, src, 0xFFFF).astype(np.uint16)
            for vector in (1, 0):
                buf = np.full(n + 48, 0xABCD, dtype=np.uint16)              # (guard words to either side)
                out = buf[16 + (start % 16):16 + (start % 16) + n]          # every 2-byte alignment of the destination
                rc = lib.east_hip_debug_narrow_symbols(src.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n,
                                                       out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), vector)
                assert rc in (0, 1), rc
                if vector:
                    vector_ran.add(rc)
                assert np.array_equal(out, want), (start, n, vector)
                assert (buf[:16 + (start % 16)] == 0xABCD).all() and (buf[16 + (start % 16) + n:] == 0xABCD).all(), (start, n, vector)
    assert lib.east_hip_debug_narrow_symbols(None, 4, None, 1) < 0
    assert vector_ran <= {0, 1}


def test_host_symbols_are_narrowed_to_bytes_when_the_text_fits_them():
    """Text whose code points all lie below 0xFF (ASCII word text: every BASELINE input) goes over the link as BYTES,
    0xFF = a terminator (east_hip.hip: upload_symbols_narrow<uint8_t>): the AVX2 form -- an alignment prologue, 32
    symbols a step through two saturating packs and a lane permutation, a tail -- and the plain loop against numpy on
    every start alignment and length, and the verdict "a symbol did not fit" (0xFF .. 0x9FF is text a byte cannot hold:
    the upload then starts over with 16-bit words) wherever such a symbol sits."""
    import ctypes
    from east import hip_backend
    lib = hip_backend.load()
    rng = np.random.default_rng(8)
    base = rng.integers(1, 0xFF, size=6000).astype(np.uint32)
    terms = np.array([0x0A00, 0x0A01, 0xFFFF, 0x10000, 0x7FFFFFFF, 0xFFFFFFFF], dtype=np.uint32)
    base[rng.integers(0, base.size, size=700)] = rng.choice(terms, size=700)
    u8p, u32p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint32)
    for start in range(0, 70):
        for n in (0, 1, 31, 32, 33, 63, 65, 100, 1000, 4096 + start):
            src = np.ascontiguousarray(base[start:start + n])
            want = np.where(src < 0xFF, src, 0xFF).astype(np.uint8)
            for vector in (1, 0):
                buf = np.full(n + 96, 0xAB, dtype=np.uint8)                 # (guard bytes to either side)
                out = buf[32 + (start % 32):32 + (start % 32) + n]          # every alignment of the destination
                rc = lib.east_hip_debug_narrow_symbols8(src.ctypes.data_as(u32p), n, out.ctypes.data_as(u8p), vector)
                assert rc & 1, (start, n, vector, rc)                       # every symbol fitted
                assert np.array_equal(out, want), (start, n, vector)
                assert (buf[:32 + (start % 32)] == 0xAB).all() and (buf[32 + (start % 32) + n:] == 0xAB).all(), (start, n, vector)
    # a text symbol a byte cannot hold, at every position of a stretch that covers prologue, vector body and tail
    src = np.ascontiguousarray(base[5:5 + 200])
    out = np.zeros(200, dtype=np.uint8)
    for bad in (0xFF, 0x100, 0x410, 0x09FF):
        for at in range(200):
            poisoned = src.copy()
            poisoned[at] = bad
            for vector in (1, 0):
                rc = lib.east_hip_debug_narrow_symbols8(poisoned.ctypes.data_as(u32p), 200, out.ctypes.data_as(u8p), vector)
                assert rc >= 0 and not (rc & 1), (bad, at, vector, rc)
    assert lib.east_hip_debug_narrow_symbols8(None, 4, None, 1) < 0


def test_bench_line_stays_under_the_drivers_tail():
    """bench.py prints ONE line for the driver, whose record keeps an 8 KB tail of stdout: the round-5 line had grown to
    21.7 KB and BENCH_r05.parsed was null.  compact_line() -- a pure function of the assembled numbers -- is run here on
    canned numbers (the full round-5 record, which holds every leg and note the bench produces, with a multi_gpu block
    added): the line must parse, stay under 6 KB, carry the contract keys, `roofline` and `cpu_baseline`, and be
    shortened rather than overflow when a leg grows."""
    import copy
    import bench
    with open(os.path.join(ROOT, "profiles", "r05_final_bench.json")) as f:
        canned = json.load(f)
    canned["multi_gpu"] = {"step_local_ms": 7.5, "allgather_ms": 0.4, "local_fraction_of_step": 0.95, "rccl_world_size": 8,
                           "backend": "nccl", "scaling_efficiency": None, "note": "x" * 400,
                           "in_process": {"step_ms": 8.1, "gather": "rccl", "rccl_ranks": 8}}
    text = bench.compact_line(canned, "/somewhere/bench_detail.json")
    assert "\n" not in text and len(text) <= bench.LINE_LIMIT == 6144
    line = json.loads(text)
    contract = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
    assert all(k in line for k in contract)
    assert line["vs_baseline"] is None and "workload" in line["config"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "avg_launch_ms",
              "algorithmic_bytes_per_launch", "next", "rocprof_hbm_fraction", "traffic_commit"):
        assert k in line["roofline"], k
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-4
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    for k in ("step_ms_min", "step_ms_median", "step_ms_max", "value_from_host", "value_first_build", "value_from_text"):
        assert k in line, k
    for leg in ("config2", "config5", "config5_prose"):
        assert {"value", "ms_per_step", "build_ms", "score_ms", "kernel"} <= set(line[leg]), leg
        assert not any(isinstance(v, (dict, list)) for v in line[leg].values())      # one-line summaries: no tables
    assert [c["n"] for c in line["worst_case"]["cases"]] == [1000, 10000, 100000]      # (the canned record is round 5's: three sizes)
    assert line["multi_gpu"]["in_process"]["rccl_ranks"] == 8 and "note" not in line["multi_gpu"]
    assert line["detail"] == "bench_detail.json"
    # values keep five significant digits of what was measured
    assert abs(line["value"] / canned["value"] - 1) < 1e-4 and abs(line["ms_per_step"] / canned["ms_per_step"] - 1) < 1e-4
    # a leg that grows is dropped (and named), the contract keys never are
    fat = copy.deepcopy(canned)
    fat["worst_case"]["cases"] = fat["worst_case"]["cases"] * 40
    short = json.loads(bench.compact_line(fat))
    assert len(json.dumps(short)) <= 6144 + 200 and "worst_case" in short["dropped_for_length"]
    assert all(k in short for k in contract)
    # no leg at all (--no-config2 --no-extras --no-cpu-baseline, or an N > 1 line): still a valid line
    bare = {k: canned[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                   "scaling", "vs_baseline", "dtype", "data", "config", "roofline")}
    assert json.loads(bench.compact_line(bare))["roofline"]["kernel"] == canned["roofline"]["kernel"]
