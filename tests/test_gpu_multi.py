"""gpu tier: several devices in one process (include/east_hip.h: east_hip_group_*, east_hip_score_table_multi;
csrc/multi.h).  The box has one GPU: the shards are LOGICAL shards on device 0 (tier 5) -- same code,
same threads, the blocks assembled by device-to-device copies --, and the RCCL path runs as a communicator of one
(ncclCommInitAll on one device).  Bar: everything identical to the single-handle build of the same collection."""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import pytest

from conftest import word_stream

pytestmark = pytest.mark.gpu

TABLES = ("suftab", "lcptab", "anntab", "childtab_up", "childtab_down", "childtab_next_l_index")


def _collection(rng, sizes):
    from east import synthetic
    docs = [synthetic.word_stream_document(rng, int(n), want_text=False)[1:] for n in sizes]
    sym = np.concatenate([d[0] for d in docs])
    off = np.concatenate([[0], np.cumsum([d[0].size for d in docs])]).astype(np.int64)
    return docs, sym, off, np.array([d[1] for d in docs], dtype=np.int32)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0]])
def test_group_of_logical_shards_equals_one_handle(hip, oracle, devices):
    from east import hip_backend, synthetic
    rng = np.random.default_rng(900 + len(devices))
    docs, sym, off, ms = _collection(rng, rng.integers(300, 60000, size=21))
    qs, qo = synthetic.keyphrases(rng, sym, 150)
    single = hip_backend.HipIndex()
    single.build(sym, off, ms)
    group = hip_backend.HipGroup(devices)
    group.build(sym, off, ms)
    assert group.first_doc.tolist() == hip_backend.shard_documents(np.diff(off), len(devices)).tolist()
    for norm in (True, False):
        assert np.array_equal(group.score_table(qs, qo, norm), single.score_table(qs, qo, norm)), norm
    info = group.info()
    assert info["shards"] == len(devices) and info["gather"] == ("rccl" if len(devices) == 1 else "copies")
    for d in (0, 5, 20):
        shard, local = group.locate(d)
        ts, tg = single.tables(d), shard.tables(local)
        for name in TABLES:
            assert np.array_equal(ts[name], tg[name]), (name, d)
    o = oracle.OracleEASA(symbols=docs[13][0], n_strings=docs[13][1])
    table = group.score_table(qs, qo, True)
    for k in range(150):
        assert table[k, 13] == o.score_symbols(qs[qo[k]:qo[k + 1]], True, fast=True)
    group.close()


def test_group_with_more_shards_than_documents(hip):
    from east import hip_backend, synthetic
    rng = np.random.default_rng(77)
    docs, sym, off, ms = _collection(rng, [4000, 900])
    qs, qo = synthetic.keyphrases(rng, sym, 40)
    single = hip_backend.HipIndex()
    single.build(sym, off, ms)
    group = hip_backend.HipGroup([0, 0, 0, 0, 0])
    group.build(sym, off, ms)
    assert sum(1 for s in group.shards if s.n_docs == 0) == 3
    assert np.array_equal(group.score_table(qs, qo, True), single.score_table(qs, qo, True))
    group.build(sym, off, ms)                                # again on the same group
    assert np.array_equal(group.score_table(qs, qo, False), single.score_table(qs, qo, False))


def test_group_from_raw_texts_and_the_measure_surface(hip):
    """east_hip_group_build_texts_v (device text preparation per shard) and MultiDeviceASTRelevanceMeasure: the same table,
    per-pair relevance, per-document views (score with per-suffix results, traversal) as the single-device measure."""
    from east import relevance, utils
    rng = np.random.default_rng(31)
    texts = [word_stream(rng, int(n)) for n in (5000, 200, 90000, 1200, 40000, 7, 3000)] + ["Ёж и ЁЛКА, снова ёж".encode("utf-8")]
    keyphrases = [word_stream(rng, int(rng.integers(5, 30))).decode() for _ in range(25)]
    prepared = [utils.prepare_text(k) for k in keyphrases]
    for normalized in (True, False):
        one = relevance.ASTRelevanceMeasure("easa", normalized, device=0)
        one.set_text_collection(texts)
        many = relevance.MultiDeviceASTRelevanceMeasure("easa", normalized, devices=[0, 0, 0])
        many.set_text_collection(texts)
        assert np.array_equal(many.relevance_table(prepared), one.relevance_table(prepared))
        assert many.relevance(prepared[3], 4) == one.relevance(prepared[3], 4)
        for d in (0, 2, 7):
            assert many.asts[d].score(prepared[1], normalized, return_suffix_scores=True) == \
                one.asts[d].score(prepared[1], normalized, return_suffix_scores=True)
    seen_one, seen_many = [], []
    one.asts[4].traverse(seen_one.append)
    many.asts[4].traverse(seen_many.append)
    assert seen_one == seen_many and len(seen_one) > 10


def test_cli_g_option_in_process(hip, tmp_path, monkeypatch):
    """`east -g 3 keyphrases table ...` with the three shards on device 0 (EAST_HIP_GROUP_DEVICES): the output of the
    single-device CLI, byte for byte; a device that does not exist is an error message and exit code 1."""
    from east import main
    rng = np.random.default_rng(8)
    tdir = tmp_path / "texts"
    tdir.mkdir()
    for i, size in enumerate([3000, 600, 5000, 1200, 2500, 40, 9000]):
        (tdir / ("doc%02d.txt" % i)).write_bytes(word_stream(rng, size))
    kp = tmp_path / "keyphrases.txt"
    kp.write_text("\n".join(word_stream(rng, int(rng.integers(6, 24))).decode() for _ in range(12)) + "\n")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "EAST_HIP_MULTI", "EAST_HIP_DEVICES", "EAST_HIP_GROUP_DEVICES"):
        monkeypatch.delenv(k, raising=False)

    def run(argv):
        buf = io.StringIO()
        with redirect_stdout(buf):
            code = main.main(argv)
        return code, buf.getvalue()

    for flags in (["-f", "csv"], ["-d"]):
        tail = flags + ["keyphrases", "table", str(kp), str(tdir)]
        want = run(tail)
        monkeypatch.setenv("EAST_HIP_GROUP_DEVICES", "0,0,0")
        assert run(["-g", "3"] + tail) == want and want[0] == 0
        monkeypatch.delenv("EAST_HIP_GROUP_DEVICES")
    n = hip.device_count()
    code, out = run(["-g", str(n + 1), "keyphrases", "table", str(kp), str(tdir)])
    assert code == 1 and "device" in out


def test_group_errors_name_the_shard(hip):
    """An error inside one shard's build or score call comes back through the group call with the shard and its device in
    the message (csrc/multi.h: on_every_shard), and the group stays usable."""
    from east import exceptions, hip_backend, synthetic
    rng = np.random.default_rng(11)
    docs, sym, off, ms = _collection(rng, [3000, 2000, 2500, 1800])
    bad = sym.copy()
    bad[off[3] - 1] = ord("A")                               # the third document no longer ends in a terminator
    group = hip_backend.HipGroup([0, 0, 0])
    with pytest.raises(exceptions.HipBackendError, match=r"shard [12] \(device 0\).*terminator"):
        group.build(bad, off, ms)
    with pytest.raises(exceptions.HipBackendError, match="no collection has been built"):
        group.score_table(np.array([65, 66], dtype=np.uint32), np.array([0, 2]))
    group.build(sym, off, ms)
    with pytest.raises(exceptions.HipBackendError, match="empty keyphrase"):
        group.score_table(np.array([65, 66], dtype=np.uint32), np.array([0, 2, 2]))
    qs, qo = synthetic.keyphrases(rng, sym, 20)
    single = hip_backend.HipIndex()
    single.build(sym, off, ms)
    assert np.array_equal(group.score_table(qs, qo, True), single.score_table(qs, qo, True))
    with pytest.raises(exceptions.HipBackendError, match="device"):
        hip_backend.HipGroup([0, hip.device_count()])


def test_distinct_handles_on_concurrent_threads(hip, oracle):
    """"A handle is not thread-safe, distinct handles are" (include/east_hip.h): six Python threads (ctypes drops the GIL),
    each with a handle of its own, build and score different collections at the same time -- raw texts through the
    streamed preparation and its uploader threads among them -- while the main thread flips a test knob back and forth
    (a call copies the knobs when it starts); every result equals the one obtained alone."""
    import threading
    from east import hip_backend, synthetic
    lib = hip.load()
    rng = np.random.default_rng(99)
    jobs = []
    for i in range(6):
        if i % 3 == 2:
            texts = [word_stream(rng, int(n)) for n in rng.integers(200000, 4000000, size=5)]
            jobs.append(("texts", texts))
        else:
            docs, sym, off, ms = _collection(rng, rng.integers(20000, 900000, size=int(rng.integers(1, 6))))
            jobs.append(("symbols", (sym, off, ms)))
    queries = synthetic.keyphrases(rng, np.concatenate([j[1][0] for j in jobs if j[0] == "symbols"]), 60)

    def run(job, out, k):
        index = hip_backend.HipIndex()
        for _ in range(3):
            if job[0] == "texts":
                index.build_texts(job[1])
            else:
                index.build(*job[1])
            table = index.score_table(queries[0], queries[1], True)
        out[k] = (table, index.tables(0, names=("suftab", "lcptab", "anntab")))

    alone, together = {}, {}
    for k, job in enumerate(jobs):
        run(job, alone, k)
    threads = [threading.Thread(target=run, args=(job, together, k)) for k, job in enumerate(jobs)]
    for t in threads:
        t.start()
    flips = 0
    while any(t.is_alive() for t in threads):
        assert lib.east_hip_debug_set_lds_rounds(flips % 2) == 0
        assert lib.east_hip_debug_set_text_stream(-1 if flips % 2 else 1 << 20) == 0
        flips += 1
    for t in threads:
        t.join()
    assert lib.east_hip_debug_set_lds_rounds(1) == 0 and lib.east_hip_debug_set_text_stream(-1) == 0
    assert len(together) == len(jobs) and flips > 0
    for k in range(len(jobs)):
        assert np.array_equal(alone[k][0], together[k][0]), k
        for name in ("suftab", "lcptab", "anntab"):
            assert np.array_equal(alone[k][1][name], together[k][1][name]), (k, name)


def _build_stub_rccl(tmp_path):
    """tests/stub_rccl.c -> a shared library (gcc; the HIP runtime is the one the process already holds)."""
    import subprocess
    from conftest import ROOT
    out = str(tmp_path / "libstub_rccl.so")
    cmd = ["gcc", "-shared", "-fPIC", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "stub_rccl.c"),
           "-o", out, "-L/opt/rocm/lib", "-lamdhip64"]
    done = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert done.returncode == 0, done.stderr.decode(errors="replace")
    return out


@pytest.mark.parametrize("n_shards", [2, 3, 8])
def test_grouped_all_gather_with_several_shards_against_a_stub_rccl(hip, tmp_path, monkeypatch, n_shards):
    """csrc/multi.h's RCCL branch -- ncclCommInitAll, then per score call ncclGroupStart, one ncclAllGather per shard with
    hipSetDevice in front of it, ncclGroupEnd, the pack kernel on the first device -- had only ever run with a group of
    ONE: RCCL refuses a communicator that names a device twice, and the box has one GPU.  Here the library is
    tests/stub_rccl.c (EAST_HIP_RCCL_LIB; device-to-device copies on the callers' streams, every call logged) and the
    mode is forced (EAST_HIP_GROUP_GATHER=rccl), so the branch runs with G = 2, 3, 8 logical shards: the table equals the
    single handle's bit for bit -- padded block maths and pack kernel included --, the communicator is created once over
    the group's device list, and every score call makes exactly one group of G all-gathers, rank 0 .. G - 1 in order, each
    of K x widest-shard doubles, each with its own shard's device current."""
    from east import hip_backend, synthetic
    lib_path = _build_stub_rccl(tmp_path)
    log = tmp_path / "calls.log"
    monkeypatch.setenv("EAST_HIP_RCCL_LIB", lib_path)
    monkeypatch.setenv("EAST_HIP_GROUP_GATHER", "rccl")
    monkeypatch.setenv("EAST_STUB_RCCL_LOG", str(log))
    rng = np.random.default_rng(1200 + n_shards)
    docs, sym, off, ms = _collection(rng, rng.integers(300, 40000, size=19))      # unequal shards: the blocks are padded
    K = 120
    qs, qo = synthetic.keyphrases(rng, sym, K)
    single = hip_backend.HipIndex()
    single.build(sym, off, ms)
    group = hip_backend.HipGroup([0] * n_shards)
    group.build(sym, off, ms)
    widths = np.diff(group.first_doc)
    assert widths.min() < widths.max()                         # (the padding is exercised)
    for norm in (True, False, True):
        assert np.array_equal(group.score_table(qs, qo, norm), single.score_table(qs, qo, norm)), norm
    assert group.info()["gather"] == "rccl"
    # fewer keyphrases, then more: the buffers are re-used / grown
    for k2 in (7, 260):
        qs2, qo2 = synthetic.keyphrases(rng, sym, k2)
        assert np.array_equal(group.score_table(qs2, qo2, True), single.score_table(qs2, qo2, True)), k2
    group.close()
    lines = log.read_text().splitlines()
    assert lines[0] == "init n=%d devices=%s" % (n_shards, ",".join("0" for _ in range(n_shards)))
    assert sum(1 for ln in lines if ln.startswith("init")) == 1            # one communicator for the group's lifetime
    assert sum(1 for ln in lines if ln.startswith("destroy")) == n_shards
    body = [ln for ln in lines[1:] if not ln.startswith("destroy")]
    per_call = n_shards + 2
    assert len(body) == 5 * per_call, len(body)
    for c, k in enumerate((K, K, K, 7, 260)):
        call = body[c * per_call:(c + 1) * per_call]
        assert call[0] == "group_start" and call[-1] == "group_end calls=%d rc=0" % n_shards, call
        for r in range(n_shards):
            assert call[1 + r] == ("allgather rank=%d count=%d dtype=8 device_at_call=0 comm_device=0 grouped=1"
                                   % (r, k * int(widths.max()))), (c, r, call[1 + r])


def test_grouped_all_gather_reports_a_failing_library(hip, tmp_path, monkeypatch):
    """An all-gather that fails inside the group (here: the stub is handed a group of three and told, through its log
    path, nothing -- the failure is a communicator destroyed under it) must come back as a HipBackendError with the
    group closed again: the next call on a fresh group works."""
    from east import exceptions, hip_backend, synthetic
    lib_path = _build_stub_rccl(tmp_path)
    monkeypatch.setenv("EAST_HIP_RCCL_LIB", lib_path)
    monkeypatch.setenv("EAST_HIP_GROUP_GATHER", "rccl")
    monkeypatch.setenv("EAST_STUB_RCCL_FAIL_AT", "1")            # the second all-gather of every group returns an error
    rng = np.random.default_rng(5)
    docs, sym, off, ms = _collection(rng, rng.integers(300, 9000, size=9))
    qs, qo = synthetic.keyphrases(rng, sym, 30)
    group = hip_backend.HipGroup([0, 0, 0])
    group.build(sym, off, ms)
    with pytest.raises(exceptions.HipBackendError) as err:
        group.score_table(qs, qo, True)
    assert "ncclAllGather failed" in str(err.value)
    monkeypatch.delenv("EAST_STUB_RCCL_FAIL_AT")
    single = hip_backend.HipIndex()
    single.build(sym, off, ms)
    assert np.array_equal(group.score_table(qs, qo, True), single.score_table(qs, qo, True))     # the same group, next call
    group.close()


def test_group_shards_with_repetitive_documents_share_the_chip(hip, oracle):
    """Every shard's build runs on a host thread of its own, and repetitive documents finish their refinement in a
    persistent kernel whose workgroups must all be resident at once (csrc/persist_rounds.h): two such launches from two
    threads must not share the chip half resident each -- the library holds a process-wide lock from the launch to the
    read-back.  Four logical shards on device 0, every document the reference's worst-case shape (identical strings) or
    copies of a passage: tables of sampled documents array_equal to the oracle, the score table equal to a single
    handle's."""
    from east import hip_backend, synthetic
    rng = np.random.default_rng(4242)
    docs = []
    for i in range(12):
        if i % 3 == 2:
            docs.append(synthetic.repeated_passage_document(rng, int(rng.integers(2000, 9000)), int(rng.integers(3, 9))))
        else:
            docs.append(synthetic.worst_case_collection(rng, int(rng.choice([20, 100])), int(rng.integers(300, 1500))))
    sym = np.concatenate([d[0] for d in docs])
    off = np.concatenate([[0], np.cumsum([d[0].size for d in docs])]).astype(np.int64)
    ms = np.array([d[1] for d in docs], dtype=np.int32)
    qs, qo = synthetic.keyphrases(rng, sym, 60)
    single = hip_backend.HipIndex()
    single.build(sym, off, ms)
    group = hip_backend.HipGroup([0, 0, 0, 0])
    for _ in range(3):                                         # (again and again: the threads meet at the lock in every order)
        group.build(sym, off, ms)
        assert sum(shard.info()["persist_rounds"] > 0 for shard in group.shards if shard.n_docs) >= 2
        assert np.array_equal(group.score_table(qs, qo, True), single.score_table(qs, qo, True))
    for d in (0, 2, 7, 11):
        shard, local = group.locate(d)
        o = oracle.OracleEASA(symbols=docs[d][0], n_strings=docs[d][1])
        t = shard.tables(local)
        for name in TABLES:
            assert np.array_equal(t[name], getattr(o, name)), (name, d)
    group.close()
