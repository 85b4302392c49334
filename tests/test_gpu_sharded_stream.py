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


def _run(world, out_dir, W, H, total, seed, *extra):
    from moving_object_detector_amd.launch import free_port
    port = free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), str(r), str(world), str(port),
                               str(out_dir), str(W), str(H), str(total), str(seed), *extra], env=env) for r in range(world)]
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


def test_two_ranks_1280x720_match_the_oracle(tmp_path, oracle):
    """BASELINE config 4's shape on the hardware at hand: a 1280x720 stream of 4 frames, reference default parameters, sharded over
    two ranks (chunk + one-plane disparity halo each) — every rank's planes, labels and objects against the ORACLE's, not only
    against a one-rank run."""
    from moving_object_detector_amd import synth
    from moving_object_detector_amd.pipeline import OBJECT_DTYPE, PLANES
    from util import bits_equal, compare_objects
    W, H, total, seed = 1280, 720, 4, 4
    two = _run(2, tmp_path, W, H, total, seed, "defaults")
    assert [(int(t["lo"]), int(t["hi"])) for t in two] == [(0, 2), (2, 4)]
    cam, s = synth.make_sequence(W, H, total, seed=seed)
    prm = synth.Params()
    found = 0
    for t in two:
        for i, f in enumerate(range(int(t["lo"]), int(t["hi"]))):
            ref = oracle.construct(cam, prm, s["disparity"][f + 1], s["disparity"][f], s["flow"][f], s["t"][f], s["q"][f], float(s["dt"][f]), "tidy")
            for j, k in enumerate(PLANES):
                assert bits_equal(t["planes"][j, i], ref[k]), (f, k)
            lab, objs, K = oracle.cluster(ref, prm, "tidy")
            assert np.array_equal(t["labels"][i], lab), f
            n = int(t["n_objects"][i])
            compare_objects(np.frombuffer(t["objects"][i, :n].tobytes(), dtype=OBJECT_DTYPE), objs, strict_velocity=True)
            found += n
    assert found > 0
