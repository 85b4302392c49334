"""Independent restatement of the reference's tracker (moving_object_tracker/src/moving_objects_tracker.cpp:14-31,54-197,
moving_object_tracker/include/kalman_tracker.hpp:17-162, kkl/include/kkl/alg/kalman_filter.hpp:62-86,
kkl/include/kkl/alg/nearest_neighbor_association.hpp:32-58, kkl/include/kkl/math/gaussian.hpp:45-71) in numpy, to check the C++ host
mirror (moving_object_detector_amd/host/moving_objects_tracker.hpp) against.  TEST INFRASTRUCTURE ONLY.  Parity unpinned: the
reference's Eigen arithmetic cannot be run here; the two restatements agree to rounding (numpy.linalg vs cofactor inverse).
The transform to the odom frame is the identity here (objects arrive in the odom frame)."""
from __future__ import annotations

import numpy as np


class Tracker:
    def __init__(self, tid, t, x):
        self.id, self.t_pred, self.t_corr, self.n = tid, t, t, 0
        self.mean = np.array(x, float)
        self.cov = np.eye(4) * 0.1
        self.Q = np.diag([0.003, 0.003, 0.01, 0.01])
        self.Rn = np.eye(4) * 0.2
        self.last = None

    def predict(self, t):
        dt = max(0.001, t - self.t_pred)
        A = np.eye(4)
        A[0, 2] = A[1, 3] = dt
        self.mean = A @ self.mean
        self.cov = A @ self.cov @ A.T + self.Q
        self.t_pred = t

    def correct(self, t, z, obj):
        K = self.cov @ np.linalg.inv(self.cov + self.Rn)
        self.mean = self.mean + K @ (z - self.mean)
        self.cov = (np.eye(4) - K) @ self.cov
        self.t_corr, self.last, self.n = t, obj, self.n + 1


def run(frames, covariance_trace_limit=0.5, correction_count_limit=3, object_radius=0.5):
    """frames: list of (stamp_seconds, [objects as (x, y, vx, vy, payload)]) -> per frame list of (id, x, y, vx, vy, payload)."""
    trackers, next_id, out = [], 0, []
    for t, objs in frames:
        for tr in trackers:
            tr.predict(t)
        pairs = []
        for i, tr in enumerate(trackers):
            Ci = np.linalg.inv(tr.cov)
            det = np.linalg.det(tr.cov)
            for j, o in enumerate(objs):
                x = np.array(o[:4], float)
                d = x - tr.mean
                sq = d @ Ci @ d
                if sq > 9.0 or np.linalg.norm(tr.mean - x) > 1.5:
                    continue
                pairs.append((-(1.0 / ((2 * np.pi) ** 2 * np.sqrt(det))) * np.exp(-0.5 * sq), i, j))
        pairs.sort(key=lambda p: p[0])
        used_t, used_o = set(), set()
        for dist, i, j in pairs:
            if i in used_t or j in used_o:
                continue
            used_t.add(i)
            used_o.add(j)
            trackers[i].correct(t, np.array(objs[j][:4], float), objs[j])
        for j, o in enumerate(objs):
            if j in used_o:
                continue
            if any(np.hypot(tr.mean[0] - o[0], tr.mean[1] - o[1]) < 2.0 * object_radius for tr in trackers):
                continue
            tr = Tracker(next_id, t, o[:4])
            tr.last = o
            next_id += 1
            trackers.append(tr)
        trackers = [tr for tr in trackers if tr.cov[0, 0] + tr.cov[1, 1] < covariance_trace_limit]
        trackers = [tr for tr in trackers if tr.cov[2, 2] + tr.cov[3, 3] < covariance_trace_limit]
        out.append(sorted((tr.id, *tr.mean, tr.last[4]) for tr in trackers if tr.n >= correction_count_limit and tr.t_corr == t))
    return out
