"""gpu tier: east.parallel over RCCL (torch.distributed backend "nccl"), one fresh child process per GPU.

The children are started with torch.distributed.run BEFORE anything in them touches a GPU (a process that
has initialised the GPU must never exec another program on this pool); this process only counts devices.
world_size 2 needs two GPUs and is skipped on the one-GPU box; world_size 1 runs everywhere and covers the
device-resident block -> all_gather_into_tensor path."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _device_count():
    import torch
    return torch.cuda.device_count()        # (counting devices does not initialise the GPU)


@pytest.mark.parametrize("world", [1, 2])
def test_nccl_table_equals_single_gpu_table(tmp_path, world):
    if _device_count() < world:
        pytest.skip("needs %d GPUs, this box has %d" % (world, _device_count()))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("EAST_HIP_DEVICE", None)
    port = 29600 + os.getpid() % 300 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "nccl_worker.py"),
           str(tmp_path)]
    done = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert done.returncode == 0, done.stdout.decode(errors="replace")[-4000:]
    singles = sorted(glob.glob(os.path.join(str(tmp_path), "single_*.npy")))
    assert len(singles) == 4
    for path in singles:
        want = np.load(path)
        key = os.path.basename(path)[len("single_"):-len(".npy")]
        for rank in range(world):
            got = np.load(os.path.join(str(tmp_path), "table_%s_%d.npy" % (key, rank)))
            assert got.shape == want.shape and np.array_equal(got, want), (key, rank)     # bit for bit, on every rank


def test_bench_distributed_line_on_one_gpu(tmp_path):
    """bench.py's N > 1 code path (RCCL process group, device-resident block into all_gather_into_tensor, the
    multi_gpu accounting) in a fresh child process on the one GPU at hand: EAST_BENCH_FORCE_DIST=1 runs the
    collective with a single rank.  The line must parse and carry the fields a scaling run is read by."""
    import json
    if _device_count() < 1:
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", EAST_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29700 + os.getpid() % 200), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    env.pop("EAST_HIP_DEVICE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--docs", "8",
           "--doc-mib", "0.25", "--keyphrases", "200", "--no-cpu-baseline", "--no-extras", "--no-config2", "--base-value", "1e9"]
    done = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert done.returncode == 0, done.stderr.decode(errors="replace")[-4000:]
    line = [ln for ln in done.stdout.decode().splitlines() if ln.startswith("{")][-1]
    assert len(line) <= 6144                                # (the driver parses the last line of an 8 KB stdout tail)
    assert line == done.stdout.decode().splitlines()[-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0 and out["scaling"] == "weak"
    mg = out["multi_gpu"]
    assert mg["rccl_world_size"] == 1 and mg["backend"] == "nccl"
    assert 0 < mg["step_local_ms"] <= out["ms_per_step"] * 1.05 and mg["allgather_ms"] >= 0
    assert 0 < mg["local_fraction_of_step"] <= 1.05
    # (the line rounds to five significant digits; bench_detail.json keeps the full figures)
    assert abs(mg["scaling_efficiency"] - out["value"] / 1e9) < 1e-3 * mg["scaling_efficiency"] and mg["scaling_base_value"] == 1e9
    # the path `east -g N` takes by default -- the in-process group, RCCL through ncclCommInitAll -- runs behind the torch
    # leg on the same shard shapes (here: a group of the one device)
    ip = mg["in_process"]
    assert "error" not in ip, ip
    assert ip["gather"] == "rccl" and ip["rccl_ranks"] == 1 and ip["shards"] == [8] and ip["step_ms"] > 0 and ip["value"] > 0
    assert out["config"]["all_gather_bytes_per_rank"] == 200 * 8 * 8
    assert out["roofline"]["kernel"] and out["roofline"]["peak"] == 8000.0


@pytest.mark.parametrize("world", [1, 2])
def test_cli_table_multi_gpu_equals_single_process(tmp_path, world):
    """`east -g N keyphrases table` (N = 2: the CLI starts its own ranks) and, on the one-GPU box, the same rank code
    under torch.distributed.run with the collective path forced (EAST_HIP_FORCE_DIST=1): the printed table is the
    single-process CLI's, byte for byte."""
    if _device_count() < world:
        pytest.skip("needs %d GPUs, this box has %d" % (world, _device_count()))
    from conftest import PKG, word_stream
    rng = np.random.default_rng(77)
    tdir = tmp_path / "texts"
    tdir.mkdir()
    for i, size in enumerate([30000, 600, 50000, 1200, 25000, 90000, 64]):
        (tdir / ("doc%02d.txt" % i)).write_bytes(word_stream(rng, size))
    kp = tmp_path / "keyphrases.txt"
    kp.write_text("\n".join(word_stream(rng, int(rng.integers(6, 24))).decode() for _ in range(40)) + "\n")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=PKG)
    for k in ("EAST_HIP_DEVICE", "EAST_HIP_DEVICES", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    tail = ["-f", "csv", "keyphrases", "table", str(kp), str(tdir)]
    single = subprocess.run([sys.executable, "-m", "east.main"] + tail, env=env, stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, timeout=600)
    assert single.returncode == 0, single.stderr.decode(errors="replace")[-3000:]
    if world == 1:
        port = 29900 + os.getpid() % 90
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
               "127.0.0.1", "--master-port", str(port), "-m", "east.main"] + tail
        env["EAST_HIP_FORCE_DIST"] = "1"
    else:
        cmd = [sys.executable, "-m", "east.main", "-g", str(world)] + tail
    table = [ln for ln in single.stdout.decode().splitlines() if ln.strip()]
    assert len(table) == 8 and table[0].count(',') == 40       # header row of 40 keyphrases + one row per text
    # -g N: the devices inside one process (the default: east_hip_score_table_multi, RCCL through ncclCommInitAll), and
    # one process per GPU under torch.distributed.run (EAST_HIP_MULTI=process)
    for mode in (("threads", "process") if world > 1 else ("process",)):
        multi = subprocess.run(cmd, env=dict(env, EAST_HIP_MULTI=mode), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert multi.returncode == 0, (mode, multi.stderr.decode(errors="replace")[-3000:])
        got = [ln for ln in multi.stdout.decode().splitlines() if ln.strip()]
        assert got[-len(table):] == table, mode     # (RCCL may print a banner in front)
