"""CPU tier: the multi-GPU path is launchable end to end.

  * `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) starts its N ranks itself, as child
    processes under torch.distributed.run, before torch is imported or a GPU touched in the parent;
  * `east -g N keyphrases table ...` / EAST_HIP_DEVICES=N does the same for the CLI;
  * under a launcher (WORLD_SIZE = 2, gloo here; RCCL on the GPU box) `east keyphrases table` builds
    DistributedASTRelevanceMeasure, every rank computes, rank 0 alone prints -- the same table as one process.

The launcher is replaced by a stub that records its command line (EAST_BENCH_LAUNCHER / EAST_HIP_LAUNCHER); the
per-shard scorer of the gloo run is the oracle-backed stand-in of test_distributed_gloo.py (the HIP scorer needs a GPU)."""
import io
import json
import os
import subprocess
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, PKG, word_stream

STUB = """
import json, os, sys
json.dump({"argv": sys.argv[1:], "world_env": os.environ.get("WORLD_SIZE"),
           "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "devices_env": os.environ.get("EAST_HIP_DEVICES")},
          open(os.environ["STUB_RECORD"], "w"))
print(json.dumps({"metric": "stub", "n_gpus": 2}))
sys.exit(7)
"""


def _stub(tmp_path):
    path = tmp_path / "stub_launcher.py"
    path.write_text(STUB)
    return "%s %s" % (sys.executable, path), str(tmp_path / "record.json")


def test_bench_gpus_2_starts_its_own_ranks(tmp_path):
    launcher, record = _stub(tmp_path)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(EAST_BENCH_LAUNCHER=launcher, STUB_RECORD=record)
    # the parent runs bench.py's __main__ and must get as far as the child's exit code without importing torch
    probe = ("import runpy, sys\n"
             "sys.argv = ['bench.py', '--gpus', '2', '--steps', '4', '--warmup', '1']\n"
             "try:\n"
             "    runpy.run_path(%r, run_name='__main__')\n"
             "except SystemExit as e:\n"
             "    code = e.code\n"
             "assert 'torch' not in sys.modules, 'the parent imported torch'\n"
             "sys.exit(code)\n" % os.path.join(ROOT, "bench.py"))
    done = subprocess.run([sys.executable, "-c", probe], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert done.returncode == 7, done.stderr.decode()[-2000:]          # the child's return code, relayed
    assert json.loads(done.stdout.decode().strip().splitlines()[-1]) == {"metric": "stub", "n_gpus": 2}   # and its line
    rec = json.load(open(record))
    argv = rec["argv"]
    assert argv[:3] == ["--nnodes=1", "--nproc-per-node", "2"]
    # (no port picked in advance: the launcher's own c10d store binds one on the loopback interface)
    assert "--rdzv-backend=c10d" in argv and "--rdzv-endpoint=127.0.0.1:0" in argv and "--master-port" not in argv
    assert argv[argv.index("--local-addr") + 1] == "127.0.0.1"
    assert argv[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"]
    assert rec["world_env"] is None and rec["ipc"] == "0"


def test_bench_gpus_n_under_a_launcher_does_not_spawn(tmp_path):
    """With WORLD_SIZE in the environment (the driver's torch.distributed.run command) parse() returns normally."""
    launcher, record = _stub(tmp_path)
    env = dict(os.environ, EAST_BENCH_LAUNCHER=launcher, STUB_RECORD=record, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    probe = ("import sys\nsys.path.insert(0, %r)\nsys.argv = ['bench.py', '--gpus', '2']\nimport bench\n"
             "args = bench.parse()\nassert args.gpus == 2\nassert 'torch' not in sys.modules\n" % ROOT)
    done = subprocess.run([sys.executable, "-c", probe], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert done.returncode == 0, done.stderr.decode()[-2000:]
    assert not os.path.exists(record)


def _write_inputs(tmp_path, n_texts=5):
    rng = np.random.default_rng(7)
    tdir = tmp_path / "texts"
    tdir.mkdir()
    for i, size in enumerate([3000, 600, 5000, 1200, 2500][:n_texts]):
        (tdir / ("doc%02d.txt" % i)).write_bytes(word_stream(rng, size))
    kp = tmp_path / "keyphrases.txt"
    kp.write_text("\n".join(word_stream(rng, int(rng.integers(6, 24))).decode() for _ in range(12)) + "\n")
    return str(kp), str(tdir)


@pytest.mark.parametrize("spelling", ["option", "env"])
def test_cli_g_option_starts_ranks(tmp_path, spelling):
    from east import main
    launcher, record = _stub(tmp_path)
    kp, tdir = _write_inputs(tmp_path)
    saved = dict(os.environ)
    try:
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            os.environ.pop(k, None)
        os.environ.update(EAST_HIP_LAUNCHER=launcher, STUB_RECORD=record, EAST_HIP_MULTI="process")    # (the launcher spelling)
        argv = ["-d", "-a", "easa", "-l", "", "-a", "easa", "-f", "csv", "keyphrases", "table", kp, tdir]
        if spelling == "option":
            argv = ["-g", "2"] + argv
        else:
            os.environ["EAST_HIP_DEVICES"] = "2"
        assert main.main(argv) == 7
    finally:
        os.environ.clear()
        os.environ.update(saved)
    rec = json.load(open(record))
    argv = rec["argv"]
    assert argv[:3] == ["--nnodes=1", "--nproc-per-node", "2"]
    tail = argv[argv.index("-m"):]
    assert tail[:2] == ["-m", "east.main"] and "-g" not in tail and tail[-4:] == ["keyphrases", "table", kp, tdir]
    assert "-d" in tail and tail[tail.index("-f") + 1] == "csv"
    assert tail[tail.index("-l") + 1] == "" and tail.count("-a") == 2      # as given: an empty value, a repeated option
    assert rec["devices_env"] is None and rec["ipc"] == "0"      # the ranks do not start ranks of their own


def _cli_rank(rank, world, port, kp, tdir, out_dir):
    for p in (PKG, ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank), EAST_HIP_DIST_BACKEND="gloo")
    from east import main
    from test_distributed_gloo import OracleMeasure
    opened = []
    real_read = main._read
    main._read = lambda path: (opened.append(path), real_read(path))[1]
    for fmt, flags in (("csv", []), ("xml", ["-d"])):
        buf = io.StringIO()
        normalized = "-d" not in flags
        real_stdout = sys.stdout
        sys.stdout = buf                                   # (what a rank prints; main() silences every rank but 0)
        try:
            code = main.main(flags + ["-f", fmt, "keyphrases", "table", kp, tdir],
                             measure_factory=lambda: OracleMeasure(normalized))
        finally:
            sys.stdout = real_stdout
        with open(os.path.join(out_dir, "out_%s_%d.txt" % (fmt, rank)), "w") as f:
            f.write("%d\n%s" % (code, buf.getvalue()))
    with open(os.path.join(out_dir, "opened_%d.json" % rank), "w") as f:
        json.dump(opened, f)


def test_cli_table_under_a_launcher_world_2_gloo(tmp_path):
    from east import applications, formatting
    from test_distributed_gloo import OracleMeasure
    kp, tdir = _write_inputs(tmp_path)
    port = 29500 + (os.getpid() % 2000) + 41
    mp.spawn(_cli_rank, args=(2, port, kp, tdir, str(tmp_path)), nprocs=2, join=True)
    keyphrases = open(kp).read().splitlines()
    texts = {name[:-4]: open(os.path.join(tdir, name), "rb").read() for name in sorted(os.listdir(tdir))}
    for fmt, normalized in (("csv", True), ("xml", False)):
        want = formatting.format_table(applications.keyphrases_table(keyphrases, texts, OracleMeasure(normalized)), fmt)
        got0 = open(os.path.join(str(tmp_path), "out_%s_0.txt" % fmt)).read()
        got1 = open(os.path.join(str(tmp_path), "out_%s_1.txt" % fmt)).read()
        assert got0 == "0\n" + want + "\n"                 # rank 0 prints the single-process table
        assert got1 == "0\n"                               # rank 1 computed, returned 0 and printed nothing
    # the documents are sharded by file size BEFORE anything is read: a rank opens the keyphrase file and its own texts
    from east import parallel
    names = sorted(os.listdir(tdir))
    shards = parallel.shard_documents([os.path.getsize(os.path.join(tdir, n)) for n in names], 2)
    for rank, (b, e) in enumerate(shards):
        opened = json.load(open(os.path.join(str(tmp_path), "opened_%d.json" % rank)))
        mine = [os.path.join(tdir, n) for n in names[b:e]]
        assert sorted(set(opened)) == sorted([kp] + mine), rank
        assert 0 < e - b < len(names)


def test_cli_g_option_runs_the_devices_in_this_process_by_default(tmp_path, monkeypatch):
    """`east -g N` without EAST_HIP_MULTI=process: no launcher, no child process -- the CLI builds the in-process
    multi-device measure (east_hip_score_table_multi) with N devices.  (The measure itself needs a GPU: gpu tier.)"""
    from east import main, relevance
    launcher, record = _stub(tmp_path)
    kp, tdir = _write_inputs(tmp_path)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "EAST_HIP_MULTI", "EAST_HIP_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("EAST_HIP_LAUNCHER", launcher)
    monkeypatch.setenv("STUB_RECORD", record)
    seen = {}

    class Recorder(object):
        def __init__(self, algorithm, normalized, devices):
            seen.update(algorithm=algorithm, normalized=normalized, devices=devices)

        def set_text_collection(self, texts, language=None):
            seen["n_texts"] = len(texts)

        def relevance_table(self, prepared):
            return np.zeros((len(prepared), seen["n_texts"]))

    monkeypatch.setattr(relevance, "MultiDeviceASTRelevanceMeasure", Recorder)
    buf = io.StringIO()
    with redirect_stdout(buf):
        assert main.main(["-g", "3", "-d", "-f", "csv", "keyphrases", "table", kp, tdir]) == 0
    assert seen == {"algorithm": "easa", "normalized": False, "devices": 3, "n_texts": 5}
    assert not os.path.exists(record)                       # the launcher was never started
    assert buf.getvalue().count("\n") == 7                  # header + five texts, print adds one
