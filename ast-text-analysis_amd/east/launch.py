# -*- coding: utf-8 -*-
"""Starting one rank per GPU without a launcher around the caller (`east -g N ...`, `python bench.py --gpus N`).

The ranks are CHILD processes under torch.distributed.run; the parent never imports torch, never touches a GPU and
never exec()s (a process that has initialised the GPU must not be replaced on this pool).  The rendezvous is
torchrun's own c10d store on a port the launcher binds itself (endpoint 127.0.0.1:0): no port is picked in
advance, so nothing can take it between the pick and the bind.
"""
import os
import shlex
import subprocess
import sys
import uuid


def launcher_command(n_ranks, target, launcher_env):
    """The command line that runs `target` (list: script path or `-m module`, then its arguments) on n_ranks ranks of
    this node.  The environment variable named launcher_env replaces the launcher itself (tests: a stub that
    records its command line)."""
    launcher = shlex.split(os.environ.get(launcher_env, "")) or [sys.executable, "-m", "torch.distributed.run"]
    return launcher + ["--nnodes=1", "--nproc-per-node", str(n_ranks), "--rdzv-backend=c10d",
                       "--rdzv-endpoint=127.0.0.1:0", "--rdzv-id", uuid.uuid4().hex, "--local-addr", "127.0.0.1"] + list(target)


def run_ranks(n_ranks, target, launcher_env, env_drop=(), pythonpath=None):
    """Runs the ranks, lets their output through (rank 0 prints) and returns their exit code."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for name in env_drop:
        env.pop(name, None)
    if pythonpath:
        env["PYTHONPATH"] = pythonpath + os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else pythonpath
    return subprocess.run(launcher_command(n_ranks, target, launcher_env), env=env).returncode
