"""Stage times when no pixel is dynamic (dynamic_speed out of reach): the fixed cost of the cluster kernels' empty tiles.
SWEEP=1: stage times over neighbor_distance x cluster_size instead."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moving_object_detector_amd import capi, synth
from moving_object_detector_amd.pipeline import Context
W, H, F, G = 1280, 720, int(os.environ.get("FRAMES", 256)), 4
cam, host = synth.make_batch(W, H, G, seed=0)
idx = [i % G for i in range(F)]
dev = torch.device("cuda:0")
ctx = Context(W, H, max_frames=F, device=0)
ctx.set_camera(cam)
import itertools
cases = [dict(dynamic_speed=0.3), dict(dynamic_speed=1e9)] if not os.environ.get("SWEEP") else \
    [dict(neighbor_distance=n, cluster_size=cs) for n, cs in itertools.product((1, 4, 8, 10), (100, 2500))]
for kw in cases:
    ctx.close()
    ctx = Context(W, H, max_frames=F, device=0, max_objects=W * H // kw.get("cluster_size", 2500) + 1)
    ctx.set_camera(cam)
    ctx.set_params(synth.Params(**kw))
    speed = kw
    ws = ctx.workspace(F)
    b = ctx.make_batch(torch.from_numpy(host["disparity_now"]).to(dev)[idx].contiguous(), torch.from_numpy(host["disparity_prev"]).to(dev)[idx].contiguous(),
                       torch.from_numpy(host["flow"]).to(dev)[idx].contiguous(), host["t"][idx], host["q"][idx], host["dt"][idx])
    for _ in range(2):
        ctx.process(b, ws)
    ctx.set_profiling(True); ctx.reset_stage_times()
    for _ in range(5):
        ctx.process(b, ws)
    ctx.synchronize()
    t = [ctx.stage_time(i) for i in range(capi.MOD_STAGE_COUNT)]
    ctx.set_profiling(False)
    print(f"dynamic_speed {speed}: us per frame", {capi.STAGE_NAMES[i]: round(1e3 * t[i][0] / t[i][1] / F, 3) for i in range(capi.MOD_STAGE_COUNT)})
