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
