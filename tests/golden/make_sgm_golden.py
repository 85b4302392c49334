#!/usr/bin/env python3
"""Generates tests/golden/sgm_320x240.npz: a synthetic stereo pair and what the SGM restatement (oracle/sgm_numpy.py) makes of
it — census words, all eight path volumes at three rows and as per-pixel sums, per-pixel sums of the total cost volume, and the
disparity map.
The estimator the reference calls (sgm_gpu) is not vendored and ships no data, so these are build-defined vectors (see
oracle/sgm_ref.cpp, "parity unpinned").  Run from the repo root:  python tests/golden/make_sgm_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sgm_numpy as sn  # noqa: E402

W, H, D = 320, 240, 128
left, right, truth = sn.make_stereo(W, H, seed=11, D=D)
st = sn.compute(left, right, D, stages=True)
rows = np.array([5, 120, 236])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sgm_320x240.npz"),
                    left=left, right=right, D=np.int32(D), P1=np.int32(6), P2=np.int32(96), rows=rows,
                    census_left=st["census_left"], census_right=st["census_right"], cost_rows=st["cost"][rows],
                    **{f"path{i}_rows": st["paths"][i][rows] for i in range(8)},
                    **{f"path{i}_sum": st["paths"][i].astype(np.uint16 if D <= 256 else np.uint32).sum(axis=2, dtype=np.uint32).astype(np.uint16)
                       for i in range(8)},       # D * 255 < 65536: a sum over the disparities fits 16 bits
                    S_sum=st["S"].astype(np.uint32).sum(axis=2), disparity=st["disparity"], truth=truth)
v = st["disparity"] >= 0
print(f"sgm_320x240: valid {v.mean():.3f}, within 1 px of the true disparity {(np.abs(st['disparity'] - truth)[v] <= 1).mean():.3f}")
