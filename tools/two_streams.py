"""Throughput of TWO contexts on two streams (256 pairs each per step) against ONE context (512 pairs per step): does the cluster stage
of one half hide behind the scene-flow kernel of the other?  python tools/two_streams.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from moving_object_detector_amd import capi, synth, pipeline

W, H, G = 1280, 720, 16
cam, sq = synth.make_sequence(W, H, G, seed=4)
dev = torch.device("cuda:0")
d = torch.from_numpy(sq["disparity"]).to(dev)
fl = torch.from_numpy(sq["flow"]).to(dev)


def make(F, stream):
    with torch.cuda.stream(stream):
        idx = [i % G for i in range(F)]
        ctx = pipeline.Context(W, H, max_frames=F)              # takes torch's CURRENT stream
        ctx.set_camera(capi.camera_struct(cam)); ctx.set_params(capi.params_struct(synth.Params()))
        ws = ctx.workspace(F)
        dn, dp, f2 = pipeline.staggered([F * H * W, F * H * W, 2 * F * H * W], torch.float32, dev)
        dn, dp, f2 = dn.view(F, H, W), dp.view(F, H, W), f2.view(F, H, W, 2)
        dn.copy_(d[1:][idx]); dp.copy_(d[:-1][idx]); f2.copy_(fl[idx])
        batch = ctx.make_batch(dn, dp, f2, sq["t"][idx], sq["q"][idx], sq["dt"][idx])
    return ctx, ws, batch


def run(parts, steps=20):
    for _ in range(3):
        for ctx, ws, b in parts:
            ctx.process(b, ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for ctx, ws, b in parts:
            ctx.process(b, ws)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


one = [make(512, torch.cuda.Stream())]
t1 = run(one)
print(f"one context, 512 pairs per step : {t1 * 1e3:.3f} ms per step, {512 / t1:.0f} pairs/s")
del one
torch.cuda.empty_cache()
two = [make(256, torch.cuda.Stream()), make(256, torch.cuda.Stream())]
t2 = run(two)
print(f"two contexts, 2 x 256 pairs     : {t2 * 1e3:.3f} ms per step, {512 / t2:.0f} pairs/s")
del two
torch.cuda.empty_cache()
four = [make(128, torch.cuda.Stream()) for _ in range(4)]
t4 = run(four)
print(f"four contexts, 4 x 128 pairs    : {t4 * 1e3:.3f} ms per step, {512 / t4:.0f} pairs/s")
