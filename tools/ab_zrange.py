"""What the cluster hand-over in the scene-flow epilogue costs the scene-flow kernel (in-process A/B on one set of buffers): the kernel as
mod_scene_flow_dev launches it (mask words only) against the kernel as mod_process_dev launches it (mask words + a tile header
per tile with a dynamic pixel + the depth range of every non-zero mask word's dynamic pixels, 8 B per 64 px = 0.125 B/px).
python tools/ab_zrange.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from moving_object_detector_amd import capi, synth, pipeline

W, H, F, G = 1280, 720, 512, 16
cam, sq = synth.make_sequence(W, H, G, seed=4)
idx = [i % G for i in range(F)]
dev = torch.device("cuda:0")
d = torch.from_numpy(sq["disparity"]).to(dev)
a, b, c = pipeline.staggered([F * H * W, F * H * W, 2 * F * H * W], torch.float32, dev)
d_now, d_prev, flow = a.view(F, H, W), b.view(F, H, W), c.view(F, H, W, 2)
d_now.copy_(d[1:][idx]); d_prev.copy_(d[:-1][idx]); flow.copy_(torch.from_numpy(sq["flow"]).to(dev)[idx])
ctx = pipeline.Context(W, H, max_frames=F)
ctx.set_camera(capi.camera_struct(cam)); ctx.set_params(capi.params_struct(synth.Params()))
ws = ctx.workspace(F)
batch = ctx.make_batch(d_now, d_prev, flow, sq["t"][idx], sq["q"][idx], sq["dt"][idx])
for _ in range(3):
    ctx.process(batch, ws); ctx.scene_flow(batch, ws)
torch.cuda.synchronize()
res = {"alone": [], "with hand-over": []}
for rep in range(8):
    for name in (("alone", "with hand-over") if rep % 2 == 0 else ("with hand-over", "alone")):
        ctx.set_profiling(True, stages=[capi.MOD_STAGE_SCENE_FLOW]); ctx.reset_stage_times()
        for _ in range(10):
            (ctx.scene_flow if name == "alone" else ctx.process)(batch, ws)
        t, n = ctx.stage_time(capi.MOD_STAGE_SCENE_FLOW)
        res[name].append(t / n)
for k, v in res.items():
    print(f"scene-flow kernel {k:15s}: " + " ".join(f"{x:.3f}" for x in v) + f" | mean {np.mean(v):.4f} ms")
print(f"hand-over costs {100 * (np.mean(res['with hand-over']) / np.mean(res['alone']) - 1):+.2f} % of the kernel")
