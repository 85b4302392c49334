"""GPU: the C++ host mirror of the reference's node / nodelet interface (moving_object_detector_amd/host/*.hpp), driven
by a standalone C++ program, against the golden fixtures."""
import os
import subprocess

import numpy as np
import pytest

from test_oracle_golden import GOLD, golden_objects, load_case
from util import PLANES, bits_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_mirror_test")


@pytest.mark.parametrize("path", [GOLD[0], GOLD[2]], ids=lambda p: os.path.basename(p))
def test_host_mirror_matches_golden(path, tmp_path):
    if not os.path.exists(EXE):
        pytest.fail("tests/cpp/host_mirror_test is not built (run __graft_entry__.build())")
    g, cam, prm = load_case(path)
    d = str(tmp_path)
    np.asarray(g["cam"], np.float64).tofile(d + "/cam.f64")
    np.asarray(g["prm"], np.float64).tofile(d + "/prm.f64")
    for k in ("d_now", "d_prev", "flow"):
        np.ascontiguousarray(g[k], np.float32).tofile(f"{d}/{k}.f32")
    np.concatenate([g["t"], g["q"], [float(np.asarray(g["dt"]).item())]]).astype(np.float64).tofile(d + "/tq.f64")
    # estimateDisparity() on the committed 320 x 240 image pair (tests/golden/sgm_320x240.npz)
    sg = np.load(os.path.join(ROOT, "tests", "golden", "sgm_320x240.npz"))
    np.asarray(sg["left"].shape[::-1], np.float64).tofile(d + "/sgm_dims.f64")
    np.ascontiguousarray(sg["left"], np.uint8).tofile(d + "/sgm_left.u8")
    np.ascontiguousarray(sg["right"], np.uint8).tofile(d + "/sgm_right.u8")
    r = subprocess.run([EXE, d], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.fromfile(d + "/sgm_disparity.f32", np.float32).reshape(sg["disparity"].shape), sg["disparity"])
    H, W = g["d_now"].shape
    cloud = np.fromfile(d + "/cloud.bin", np.float32).reshape(H, W, 8)
    # dt reaches the library the way ros::Duration::toSec() forms it from (sec, nsec) stamps: exact for the fixtures' dt = 0.1
    dt = float(np.asarray(g["dt"]).item())
    nsec = int(np.floor((100.25 + dt - np.floor(100.25 + dt)) * 1e9 + 0.5)) - 250000000
    sec = int(np.floor(100.25 + dt)) - 100
    if nsec < 0:
        sec, nsec = sec - 1, nsec + 1000000000
    assert float(sec) + 1e-9 * float(nsec) == dt, "fixture dt must survive the (sec, nsec) round trip"
    for j, k in zip((0, 1, 2, 4, 5, 6), PLANES):
        assert bits_equal(cloud[..., j], g[k]), k
    labels = np.fromfile(d + "/labels.i32", np.int32).reshape(H, W)
    assert np.array_equal(labels, g["labels"])
    objs = np.fromfile(d + "/objects.f64", np.float64).reshape(-1, 14)
    want = golden_objects(g)
    assert len(objs) == len(want)
    for o, e in zip(objs, want):
        assert int(o[0]) == e["id"] and np.array_equal(o[4:8], [0, 0, 0, 1])
        assert np.array_equal(o[1:4], e["center"]) and np.array_equal(o[11:14], e["bounding_box"])
        # velocity = the cluster's median member (clusterer_nodelet.cpp:168-174), widened to F64 in the message
        if not e["ambiguous"]:
            assert np.array_equal(o[8:11], np.asarray(e["velocity"], np.float64)), (o[8:11], e["velocity"])
        else:   # a tie between different vectors: the fixture (numpy, no std::sort) only pins the norm
            norm = lambda v: np.float32(np.sqrt(np.float32(v[0]) ** 2 + (np.float32(v[1]) ** 2 + np.float32(v[2]) ** 2)))
            assert norm(o[8:11]) == norm(e["velocity"])
