#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU: 1920x1080 ZED-style stream, end to end including the on-GPU disparity estimator.

A step = one batch of F new stereo image pairs: mod_sgm_compute_dev (F disparity planes, written behind the plane of the frame
before them: previous = D[0..F-1], now = D[1..F]) + mod_process_dev (scene flow + clustering).  Images, flow and ego-motion are
synthetic and HBM-resident; the optical flow is not estimated (the reference's estimator is a Caffe network, out of scope).
Prints one JSON line.  usage (GPU box): python tools/bench_config5.py [--frames 64] [--steps 5] [--width 1920 --height 1080]"""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moving_object_detector_amd import capi, synth
from moving_object_detector_amd.pipeline import Context

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=64); ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--distinct", type=int, default=8); ap.add_argument("--disparities", type=int, default=128); ap.add_argument("--paths", type=int, default=8)
a = ap.parse_args()
W, H, F, G, D = a.width, a.height, a.frames, min(a.distinct, a.frames), a.disparities
pairs = [synth.make_stereo_images(W, H, 100 + k, D, n_boxes=5) for k in range(G)]
cam = synth.make_camera(W, H)
cam.min_disparity, cam.max_disparity = np.float32(0.0), np.float32(D - 1)
ctx = Context(W, H, max_frames=F)
ctx.set_camera(cam); ctx.set_params(synth.Params())
dev = ctx.device
idx = [i % G for i in range(F)]
left = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)[idx].contiguous()
right = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)[idx].contiguous()
rng = np.random.default_rng(1)
flow = torch.from_numpy(rng.uniform(-3, 3, size=(G, H, W, 2)).astype(np.float32)).to(dev)[idx].contiguous()
ts = np.tile(np.array([[0.01, 0.0, 0.08]]), (F, 1)); qs = np.tile(np.array([[0.0, 0.003, 0.0, 1.0]]), (F, 1)); dts = np.full(F, 1.0 / 15.0)
disp = torch.zeros((F + 1, H, W), dtype=torch.float32, device=dev)
sp = capi.ModSgmParams(D, 6, 96, a.paths, 1, 1)
ws = ctx.workspace(F)
batch = ctx.make_batch(disp[1:], disp[:-1], flow, ts, qs, dts)
assert ctx.lib.mod_sgm_compute_dev(ctx.h, 1, left.data_ptr(), right.data_ptr(), C.byref(sp), disp.data_ptr()) == 0     # the stream's first plane

def step():
    assert ctx.lib.mod_sgm_compute_dev(ctx.h, F, left.data_ptr(), right.data_ptr(), C.byref(sp), disp[1:].data_ptr()) == 0
    assert ctx.process(batch, ws) == 0

for _ in range(a.warmup): step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
t0 = time.perf_counter()
sgm_ms = 0.0
for _ in range(a.steps):
    ev[0].record()
    ctx.lib.mod_sgm_compute_dev(ctx.h, F, left.data_ptr(), right.data_ptr(), C.byref(sp), disp[1:].data_ptr())
    ev[1].record()
    ctx.process(batch, ws)
    ev[2].record()
    torch.cuda.synchronize()
    sgm_ms += ev[0].elapsed_time(ev[1])
el = time.perf_counter() - t0
print(json.dumps({"metric": f"stereo pairs/sec end to end incl. on-GPU SGM disparity at {W}x{H}", "value": F * a.steps / el, "unit": "stereo pairs/s",
                  "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * el / a.steps, "ms_per_frame": 1e3 * el / (a.steps * F),
                  "sgm_ms_per_frame": sgm_ms / (a.steps * F), "scene_flow_and_cluster_ms_per_frame": 1e3 * el / (a.steps * F) - sgm_ms / (a.steps * F),
                  "data": "synthetic", "dtype": "u8/u16 (disparity) + f32+f64 (scene flow)",
                  "config": {"workload": f"{W}x{H} synthetic stereo images -> SGM disparity (D={D}, {a.paths} paths, median, left-right check) -> scene flow + clusters, "
                             f"HBM-resident, synthetic flow / ego-motion", "frames_per_step": F, "distinct_frames": G},
                  "objects_per_frame_mean": float(ws["n_objects"].float().mean()), "valid_disparity_share": float((disp[1:] >= 0).float().mean())}))
ctx.close()
