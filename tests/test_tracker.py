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


def test_kalman_states_against_exact_rational_arithmetic(exe):
    """Known answers that do NOT come from this repository's numpy restatement: one object seen in eight frames; the filter of
    kalman_tracker.hpp:31-50 (transition I with dt at (0,2),(1,3); process noise diag(0.003, 0.003, 0.01, 0.01); measurement = I,
    noise 0.2 I; initial covariance 0.1 I) and kalman_filter.hpp:62-86 (predict, correct with the full 4 x 4 gain) is evaluated in
    EXACT rational arithmetic (the double constants taken at their binary values) and the mirror's reported states must agree to
    1e-12 relative — rounding only."""
    from fractions import Fraction as Fr

    def mat(n, m, f=lambda i, j: Fr(0)):
        return [[f(i, j) for j in range(m)] for i in range(n)]

    def mul(A, B):
        return [[sum(A[i][k] * B[k][j] for k in range(len(B))) for j in range(len(B[0]))] for i in range(len(A))]

    def add(A, B):
        return [[A[i][j] + B[i][j] for j in range(len(A[0]))] for i in range(len(A))]

    def tr(A):
        return [list(r) for r in zip(*A)]

    def inv(A):                                                  # Gauss-Jordan on rationals
        n = len(A)
        M = [list(A[i]) + [Fr(int(i == j)) for j in range(n)] for i in range(n)]
        for c in range(n):
            p = next(r for r in range(c, n) if M[r][c] != 0)
            M[c], M[p] = M[p], M[c]
            M[c] = [v / M[c][c] for v in M[c]]
            for r in range(n):
                if r != c and M[r][c] != 0:
                    M[r] = [a - M[r][c] * b for a, b in zip(M[r], M[c])]
        return [row[n:] for row in M]

    stamps = [(200, 0), (200, 66666667), (200, 133333334), (200, 200000001), (200, 266666668), (200, 333333335), (200, 400000002), (200, 466666669)]
    obs = [(1.0 + 0.05 * k, -2.0 + 0.02 * k, 0.75, 0.3) for k in range(len(stamps))]
    text = f"{len(stamps)}\n" + "".join(f"{s} {ns} 1\n" + "%.17g %.17g %.17g %.17g %.17g\n" % (o + (float(k),)) for k, ((s, ns), o) in enumerate(zip(stamps, obs)))
    r = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    I4 = mat(4, 4, lambda i, j: Fr(int(i == j)))
    Rn = mat(4, 4, lambda i, j: (Fr(0.003) if i < 2 else Fr(0.01)) if i == j else Fr(0))
    Qn = mat(4, 4, lambda i, j: Fr(0.2) if i == j else Fr(0))
    mean = [[Fr(v)] for v in obs[0]]                                 # the tracker is born on the first detection (moving_objects_tracker.cpp:186-196)
    cov = mat(4, 4, lambda i, j: Fr(0.1) if i == j else Fr(0))
    reported = 0
    for k in range(1, len(stamps)):
        dsec, dnsec = stamps[k][0] - stamps[k - 1][0], stamps[k][1] - stamps[k - 1][1]
        dt = max(0.001, float(dsec) + 1e-9 * float(dnsec))           # ros::Duration::toSec(), then std::max(0.001, .) (kalman_tracker.hpp:66-67)
        A = [list(row) for row in I4]
        A[0][2] = A[1][3] = Fr(dt)
        mean = mul(A, mean)
        cov = add(mul(mul(A, cov), tr(A)), Rn)
        K = mul(cov, inv(add(cov, Qn)))                               # C = I
        z = [[Fr(v)] for v in obs[k]]
        mean = add(mean, mul(K, [[z[i][0] - mean[i][0]] for i in range(4)]))
        cov = mul([[I4[i][j] - K[i][j] for j in range(4)] for i in range(4)], cov)
        v = lines[k].split()
        if k >= 3:                                                    # reported from the third correction on (correction_count_limit = 3)
            assert int(v[0]) == 1, (k, lines[k])
            got = [float(x) for x in v[2:6]]
            want = [float(mean[i][0]) for i in range(4)]
            assert np.allclose(got, want, rtol=1e-12, atol=1e-14), (k, got, want)
            assert float(v[6]) == float(k)                            # the payload of the detection associated last
            reported += 1
        else:
            assert int(v[0]) == 0, (k, lines[k])
    assert reported == len(stamps) - 3
