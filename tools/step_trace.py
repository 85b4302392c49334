"""Per-step times of a long run (torch events on the context's stream between the steps): how a step's time develops over the first
hundreds of steps, cluster stage in one piece vs in chunks.  python tools/step_trace.py [steps] [setting ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from moving_object_detector_amd import capi, synth, pipeline

W, H, F, G = 1280, 720, 512, 16
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
settings = [int(a, 0) for a in sys.argv[2:]] or [1, 2]
cam, sq = synth.make_sequence(W, H, G, seed=4)
idx = [i % G for i in range(F)]
dev = torch.device("cuda:0")
d = torch.from_numpy(sq["disparity"]).to(dev)
a, b, c = pipeline.staggered([F * H * W, F * H * W, 2 * F * H * W], torch.float32, dev)
d_now, d_prev, flow = a.view(F, H, W), b.view(F, H, W), c.view(F, H, W, 2)
d_now.copy_(d[1:][idx]); d_prev.copy_(d[:-1][idx]); flow.copy_(torch.from_numpy(sq["flow"]).to(dev)[idx])
ts, qs, dts = sq["t"][idx], sq["q"][idx], sq["dt"][idx]
for st in settings:
    ctx = pipeline.Context(W, H, max_frames=F, batch_chunks=st)
    ctx.set_camera(capi.camera_struct(cam)); ctx.set_params(capi.params_struct(synth.Params()))
    ws = ctx.workspace(F)
    batch = ctx.make_batch(d_now, d_prev, flow, ts, qs, dts)
    torch.cuda.synchronize()
    time.sleep(2.0)                                    # the GPU idles before the run, as before a fresh process's first step
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        ctx.process(batch, ws)
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])
    print(f"setting {st:#x}: first 30 steps: " + " ".join(f"{x:.2f}" for x in t[:30]), flush=True)
    print(f"setting {st:#x}: ms per step, means of consecutive tens: " + " ".join(f"{t[i:i + 10].mean():.3f}" for i in range(0, steps, 10)), flush=True)
    ctx.close()
    del ctx, ws, batch
    torch.cuda.empty_cache()
