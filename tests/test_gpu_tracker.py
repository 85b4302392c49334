"""GPU: SURVEY.md section 8(f) row 4 end to end — a 10-frame 640 x 480 stream goes through the HIP path (mod_process_dev), the
GPU's ModObjects go through the C++ tracker mirror (moving_object_detector_amd/host/moving_objects_tracker.hpp, driven by
tests/cpp/tracker_test.cpp; transform to odom = identity, output frame "odom" checked inside the driver) and the tracked objects are
compared with the ORACLE's objects run through oracle/tracker_numpy.py.  Reference: moving_object_tracker/src/
moving_objects_tracker.cpp:54-197, kkl/include/kkl/alg/kalman_filter.hpp:62-86.  Floating point on the tracker side: 1e-9 relative
(written here); ids, counts and the payload of the last associated detection exact."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpu_objects_through_the_tracker_mirror_match_the_oracle_chain(tmp_path):
    from moving_object_detector_amd import synth
    from moving_object_detector_amd.pipeline import Context
    from oracle import pyoracle, tracker_numpy
    exe = str(tmp_path / "tracker_test")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "tracker_test.cpp"), "-o", exe])
    W, H, F = 640, 480, 10
    cam, sq = synth.make_sequence(W, H, F, seed=2)
    prm = synth.Params(cluster_size=600)
    ctx = Context(W, H, max_frames=F)
    ctx.set_camera(cam); ctx.set_params(prm)
    ws = ctx.workspace(F)
    d = torch.from_numpy(sq["disparity"]).to(ctx.device)
    batch = ctx.make_batch(d[1:].contiguous(), d[:-1].contiguous(), torch.from_numpy(sq["flow"]).to(ctx.device), sq["t"], sq["q"], sq["dt"])
    assert ctx.process(batch, ws) == 0
    ctx.synchronize()
    gpu_objs = ctx.objects_to_host(ws)
    ctx.close()
    stamps = [(100 + (f // 10), (f % 10) * 100000000) for f in range(F)]          # dt = 0.1 s, exact in (sec, nsec)
    rec = lambda c, v, b: (float(c[0]), float(c[1]), float(v[0]), float(v[1]), float(b[0]))
    text, chain = f"{F}\n", []
    for f in range(F):
        g = [rec(o["center"], o["velocity"], o["bounding_box"]) for o in gpu_objs[f]]
        text += f"{stamps[f][0]} {stamps[f][1]} {len(g)}\n" + "".join("%.17g %.17g %.17g %.17g %.17g\n" % q for q in g)
        ref = pyoracle.construct(cam, prm, sq["disparity"][f + 1], sq["disparity"][f], sq["flow"][f], sq["t"][f], sq["q"][f], float(sq["dt"][f]), "tidy")
        _, objs, _ = pyoracle.cluster(ref, prm, "tidy", max_objects=W * H)
        chain.append((stamps[f][0] + 1e-9 * stamps[f][1], [rec(o["center"], o["velocity"], o["bounding_box"]) for o in objs]))
    r = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr                                           # (4 = output frame is not "odom" / stamp not the input's)
    want = tracker_numpy.run(chain)
    lines = r.stdout.strip().splitlines()
    assert len(lines) == F
    tracked = 0
    for f, (line, w) in enumerate(zip(lines, want)):
        v = line.split()
        assert int(v[0]) == len(w), (f, v[0], len(w))
        for k in range(len(w)):
            got = v[1 + 6 * k: 7 + 6 * k]
            assert int(got[0]) == w[k][0], (f, k)
            assert np.allclose([float(x) for x in got[1:5]], w[k][1:5], rtol=1e-9, atol=1e-12), (f, k)
            assert float(got[5]) == w[k][5]
        tracked += len(w)
    assert tracked >= 5                                                          # tracks do get confirmed on this stream
