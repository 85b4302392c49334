#!/usr/bin/env python3
"""Generates the committed golden fixtures (inputs + expected outputs) for the scene-flow / clustering path.

The reference ships no tests or data for this path, so the fixtures are produced by the *numpy*
restatement (oracle/numpy_ref.py) from seeded synthetic inputs and hand-built edge cases; the C++ oracle and the HIP
path are then both checked against them.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moving_object_detector_amd import synth  # noqa: E402
from oracle import numpy_ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def run_case(name, cam, prm, d_now, d_prev, flow, t, q, dt):
    sf = numpy_ref.scene_flow(cam, prm, d_now, d_prev, flow, t, q, dt)
    labels, objs, K = numpy_ref.cluster(prm, *[sf[k] for k in ("x", "y", "z", "vx", "vy", "vz")])
    out = {
        "cam": np.array([cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.Tx, cam.Ty, cam.disp_f, cam.disp_T,
                         cam.min_disparity, cam.max_disparity], np.float64),
        "prm": np.array([prm.dynamic_flow_diff, prm.cluster_size, prm.neighbor_distance, prm.depth_diff, prm.dynamic_speed], np.float64),
        "d_now": d_now.astype(np.float32), "d_prev": d_prev.astype(np.float32), "flow": flow.astype(np.float32),
        "t": np.asarray(t, np.float64), "q": np.asarray(q, np.float64), "dt": np.float64(dt),
        "labels": labels.astype(np.int32), "K": np.int32(K), "n_objects": np.int32(len(objs)),
        "static_flow": sf["static_flow"], "depth": sf["depth"],
    }
    for k in ("x", "y", "z", "vx", "vy", "vz"):
        out[k] = sf[k]
    if objs:
        out["obj_n_points"] = np.array([o["n_points"] for o in objs], np.int32)
        out["obj_center"] = np.stack([o["center"] for o in objs])
        out["obj_bbox"] = np.stack([o["bounding_box"] for o in objs])
        out["obj_velocity"] = np.stack([o["velocity"] for o in objs])
        out["obj_ambiguous"] = np.array([o["ambiguous"] for o in objs], np.int32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: {cam.width}x{cam.height} K={K} objects={len(objs)} dynamic={int(numpy_ref.dynamic_mask(prm, sf['vx'], sf['vy'], sf['vz']).sum())}")


def synthetic(name, W, H, seed, frame, **prm_kw):
    cam, f = synth.make_frame(W, H, seed=seed, frame=frame)
    run_case(name, cam, synth.Params(**prm_kw), f.disparity_now, f.disparity_prev, f.flow, f.translation, f.quaternion, f.dt)


def edge_cases():
    """Hand-built 48x32 frame hitting every branch listed in SURVEY.md §4/§8(c): NaN/inf/negative/zero/out-of-range
    disparity, NaN and huge flow, out-of-image warp, sub-threshold residual, singleton dynamic pixel, depth-gated
    neighbours, the up-left-only window asymmetry, a sub-cluster_size cluster."""
    W, H = 48, 32
    cam = synth.make_camera(W, H)
    cam.fx = cam.fy = 50.0
    cam.disp_f = np.float32(50.0)
    cam.min_disparity = np.float32(-4.0)            # lets a negative disparity through getDisparity's range gate
    prm = synth.Params(dynamic_flow_diff=1, cluster_size=6, neighbor_distance=2, depth_diff=0.15, dynamic_speed=0.3)
    d_now = np.full((H, W), 2.0, np.float32)          # z = 50*0.12/2 = 3 m
    d_prev = np.full((H, W), 2.0, np.float32)
    flow = np.zeros((H, W, 2), np.float32)
    t = np.zeros(3)
    q = np.array([0.0, 0.0, 0.0, 1.0])
    # invalid disparity classes (now and prev)
    d_now[0, 0] = np.nan; d_now[0, 1] = np.inf; d_now[0, 2] = -1.0; d_now[0, 3] = 0.0; d_now[0, 4] = 200.0; d_now[0, 5] = 1e-42
    d_prev[1, 0] = np.nan; d_prev[1, 1] = np.inf; d_prev[1, 2] = -1.0; d_prev[1, 3] = 0.0; d_prev[1, 4] = 200.0
    # flow classes
    flow[2, 0] = (np.nan, 0); flow[2, 1] = (0, np.nan); flow[2, 2] = (1e6, 0); flow[2, 3] = (-1e6, 0); flow[2, 4] = (np.inf, 0)
    flow[2, 5] = (3e9, 0); flow[2, 6] = (0.5, 0.5); flow[2, 7] = (-0.5, -0.5); flow[2, 8] = (1.5, 2.5)   # rounding half away
    flow[3, 0] = (5.0, 0)                                  # warps out of the image on the left
    flow[3, W - 1] = (-5.0, 0)                             # ... on the right
    flow[4, 10] = (0.4, 0.3)                               # residual below dynamic_flow_diff -> v = 0
    # moving block A (6x5) with a depth step in the middle: split by the depth gate
    flow[8:13, 10:16, 0] = 3.0
    d_prev[8:13, 7:13] = 2.2                               # where the block came from: different depth -> velocity
    d_now[8:13, 13:16] = 1.9                               # depth step > 0.15 m inside the block
    # moving block B (4x3 = 12 px), connected only through an up-RIGHT diagonal to block C: must stay separate
    flow[20:23, 20:24, 0] = 3.0
    d_prev[20:23, 17:21] = 2.3
    flow[17:20, 24:28, 0] = 3.0                            # block C sits up-right of B
    d_prev[17:20, 21:25] = 2.3
    # singleton dynamic pixel and a 2-pixel cluster below cluster_size
    flow[28, 5, 0] = 4.0; d_prev[28, 1] = 2.5
    flow[28, 30, 0] = 4.0; flow[28, 31, 0] = 4.0; d_prev[28, 26:28] = 2.5
    # gap bridged by the window (n = 2): two 3x3 blocks one empty column apart
    flow[26:29, 38:41, 0] = 3.0; flow[26:29, 42:45, 0] = 3.0; d_prev[26:29, 35:42] = 2.4
    run_case("edge_48x32", cam, prm, d_now, d_prev, flow, t, q, 0.1)


if __name__ == "__main__":
    edge_cases()
    synthetic("synth_64x48", 64, 48, seed=21, frame=0, dynamic_flow_diff=1, cluster_size=20)
    synthetic("synth_160x120", 160, 120, seed=22, frame=1, dynamic_flow_diff=1, cluster_size=60)
    synthetic("synth_160x120_n10", 160, 120, seed=23, frame=0, dynamic_flow_diff=1, cluster_size=40, neighbor_distance=10)
    synthetic("synth_320x240", 320, 240, seed=24, frame=2, dynamic_flow_diff=2, cluster_size=150, neighbor_distance=1)
