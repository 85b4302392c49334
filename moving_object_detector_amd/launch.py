"""One process per GPU: start N ranks of a script as fresh child processes.

The parent must not have touched the GPU (no HIP call, no ``torch.cuda.is_available()``): it only picks a rendezvous port,
starts ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`` — the very command
the round driver uses — and hands back the exit code.  The children inherit stdout, so rank 0's result line goes straight
through.  Nothing is ever exec'ed over a process that holds the device.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_command(script: str, argv: list, n: int, port: int) -> list:
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script] + list(argv)


def spawn_ranks(script: str, argv: list, n: int, port: int | None = None, timeout: float | None = None) -> int:
    """Run `script argv` as n ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set by torch.distributed.run); returns the exit code."""
    if "WORLD_SIZE" in os.environ:
        raise RuntimeError("spawn_ranks called from inside a rank")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = rank_command(script, argv, n, port or free_port())
    proc = subprocess.Popen(cmd, env=env)
    try:
        return proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.terminate()                                   # the exact child we started; torch.distributed.run ends its ranks
        try:
            proc.wait(timeout=30)
        except subprocess.TimeoutExpired:
            proc.kill()
        return 124
