"""CPU: the C++ host mirror of the reference's tracker (Kalman constant-velocity tracks, gated
nearest-neighbour association, covariance pruning) against the numpy restatement on scripted object streams.  Floating point,
Eigen not available on either side: tolerance 1e-9 relative (written here), ids and track sets exact."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stream(seed, frames=60, n_obj=5):
    """Objects on straight lines with measurement noise; some vanish for a while, some appear late, some cross."""
    rng = np.random.default_rng(seed)
    p0 = rng.uniform(-10, 10, size=(n_obj, 2))
    v = rng.uniform(-1.5, 1.5, size=(n_obj, 2))
    born = rng.integers(0, 15, size=n_obj)
    gap = [(int(a), int(a) + int(rng.integers(2, 8))) for a in rng.integers(20, 45, size=n_obj)]
    out = []
    for f in range(frames):
        t = 100.0 + f / 15.0
        sec, nsec = int(t), int(round((t - int(t)) * 1e9))
        objs = []
        for k in range(n_obj):
            if f < born[k] or gap[k][0] <= f < gap[k][1]:
                continue
            pos = p0[k] + v[k] * (f / 15.0) + rng.normal(0, 0.03, 2)
            vel = v[k] + rng.normal(0, 0.05, 2)
            objs.append((float(pos[0]), float(pos[1]), float(vel[0]), float(vel[1]), float(100 * f + k)))
        if f % 17 == 3:                                          # a spurious detection far away
            objs.append((50.0 + f, -40.0, 0.0, 0.0, float(100 * f + 99)))
        out.append((sec, nsec, objs))
    return out


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    p = str(tmp_path_factory.mktemp("trk") / "tracker_test")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "tracker_test.cpp"), "-o", p])
    return p


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_host_mirror_matches_numpy_restatement(exe, seed):
    from oracle import tracker_numpy
    stream = _stream(seed)
    text = f"{len(stream)}\n" + "".join(f"{s} {ns} {len(o)}\n" + "".join("%.17g %.17g %.17g %.17g %.17g\n" % q for q in o) for s, ns, o in stream)
    r = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    want = tracker_numpy.run([(s + 1e-9 * ns, o) for s, ns, o in stream])
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(stream)
    tracked_total = 0
    for f, (line, w) in enumerate(zip(lines, want)):
        v = line.split()
        n = int(v[0])
        assert n == len(w), (f, n, len(w))
        for k in range(n):
            rec = v[1 + 6 * k: 7 + 6 * k]
            assert int(rec[0]) == w[k][0], (f, k)
            got = np.array([float(x) for x in rec[1:5]])
            assert np.allclose(got, np.array(w[k][1:5]), rtol=1e-9, atol=1e-12), (f, k, got, w[k])
            assert float(rec[5]) == w[k][5]                    # the payload of the LAST associated detection
        tracked_total += n
    assert tracked_total > 60                                   # tracks do get confirmed (>= 3 corrections) and reported
