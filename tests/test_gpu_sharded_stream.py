"""GPU: a stream sharded over two processes gives the bytes of a single-process run.

Two fresh child processes (gloo rendezvous, both on cuda:0 — a one-GPU box) each take a contiguous chunk of one 8-frame
sequence plus the one-plane disparity halo, run the HIP path and save their outputs; the concatenation must equal the
single-process result byte for byte: planes, labels, object counts and object records."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(world, out_dir, W, H, total, seed):
    from moving_object_detector_amd.launch import free_port
    port = free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), str(r), str(world), str(port),
                               str(out_dir), str(W), str(H), str(total), str(seed)], env=env) for r in range(world)]
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=600))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(-9)
    assert rcs == [0] * world, rcs
    return [np.load(os.path.join(out_dir, f"rank{r}of{world}.npz")) for r in range(world)]


@pytest.mark.parametrize("total", [8, 7])
def test_two_ranks_equal_one(tmp_path, total):
    W, H, seed = 320, 240, 21
    one = _run(1, tmp_path, W, H, total, seed)[0]
    two = _run(2, tmp_path, W, H, total, seed)
    assert [(int(t["lo"]), int(t["hi"])) for t in two] == [(0, (total + 1) // 2), ((total + 1) // 2, total)]
    planes = np.concatenate([t["planes"] for t in two], axis=1)
    assert planes.tobytes() == one["planes"].tobytes()              # byte for byte, NaN payloads included
    assert np.array_equal(np.concatenate([t["labels"] for t in two]), one["labels"])
    n1 = one["n_objects"]
    n2 = np.concatenate([t["n_objects"] for t in two])
    assert np.array_equal(n1, n2) and n1.sum() > 0                  # the stream does contain objects
    o2 = np.concatenate([t["objects"] for t in two])
    for f in range(total):
        assert o2[f, :n1[f]].tobytes() == one["objects"][f, :n1[f]].tobytes()
