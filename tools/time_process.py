"""Wall-clock pairs/s of Context.process with the stage timers off (compare MOD_OVERLAP=0/1)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moving_object_detector_amd import capi, synth
from moving_object_detector_amd.pipeline import Context

W, H, F, G = 1280, 720, int(os.environ.get("FRAMES", 64)), 4
cam, host = synth.make_batch(W, H, G, seed=0)
idx = [i % G for i in range(F)]
dev = torch.device("cuda:0")
d_now = torch.from_numpy(host["disparity_now"]).to(dev)[idx].contiguous()
d_prev = torch.from_numpy(host["disparity_prev"]).to(dev)[idx].contiguous()
flow = torch.from_numpy(host["flow"]).to(dev)[idx].contiguous()
ctx = Context(W, H, max_frames=F, device=0)
ctx.set_camera(capi.camera_struct(synth.make_camera(W, H)))
ctx.set_params(capi.params_struct(synth.Params()))
ws = ctx.workspace(F)
batch = ctx.make_batch(d_now, d_prev, flow, host["t"][idx], host["q"][idx], host["dt"][idx])
for _ in range(3):
    ctx.process(batch, ws)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    ctx.process(batch, ws)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"overlap={os.environ.get('MOD_OVERLAP', '1')} frames={F} pairs/s {F * n / dt:.0f}  ms/launch {dt / n * 1e3:.3f}")
