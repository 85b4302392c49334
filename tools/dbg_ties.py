import ctypes as C, os, sys
os.environ["MOD_DEBUG"] = "0"   # the tile kernel's phase counters (bit 128) share slots with the tie kernel's
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moving_object_detector_amd import synth
from moving_object_detector_amd.pipeline import Context
W, H = 1280, 720
cam, host = synth.make_batch(W, H, 1, seed=0, first_frame=3)
ctx = Context(W, H, max_frames=1); ctx.set_camera(cam); ctx.set_params(synth.Params()); ws = ctx.workspace(1)
dev = ctx.device
b = ctx.make_batch(torch.from_numpy(host["disparity_now"]).to(dev), torch.from_numpy(host["disparity_prev"]).to(dev), torch.from_numpy(host["flow"]).to(dev), host["t"], host["q"], host["dt"])
lib = ctx.lib; lib.mod_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
out = (C.c_uint64 * 96)()
for it in range(3):
    ctx.process(b, ws); ctx.synchronize(); lib.mod_debug_counters(ctx.h, out)
names = ["bbox", "count", "fill+keys", "-", "hbm levels", "lds levels+finish"]
for i, n in enumerate(names): print(f"{n:18s} {out[20+i]:10d} x10ns")
