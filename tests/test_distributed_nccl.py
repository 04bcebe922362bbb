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
           "--doc-mib", "0.25", "--keyphrases", "200", "--no-cpu-baseline", "--no-extras", "--no-config2"]
    done = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert done.returncode == 0, done.stderr.decode(errors="replace")[-4000:]
    line = [ln for ln in done.stdout.decode().splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0 and out["scaling"] == "weak"
    mg = out["multi_gpu"]
    assert mg["rccl_world_size"] == 1 and mg["backend"] == "nccl"
    assert 0 < mg["step_local_ms"] <= out["ms_per_step"] * 1.05 and mg["allgather_ms"] >= 0
    assert 0 < mg["weak_scaling_efficiency"] <= 1.05
    assert out["config"]["all_gather_bytes_per_rank"] == 200 * 8 * 8
    assert out["roofline"]["kernel"] and out["roofline"]["peak"] == 8000.0
