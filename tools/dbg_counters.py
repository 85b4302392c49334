"""Phase cycle counters of k_ccl_tile (MOD_DEBUG=128). Diagnostic only."""
import ctypes as C, os, sys
os.environ["MOD_DEBUG"] = os.environ.get("MOD_DEBUG", "128")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moving_object_detector_amd import synth, capi
from moving_object_detector_amd.pipeline import Context
W, H, F = 1280, 720, 16
cam, host = synth.make_batch(W, H, 4, seed=0)
idx = [i % 4 for i in range(F)]
ctx = Context(W, H, max_frames=F)
ctx.set_camera(cam); ctx.set_params(synth.Params())
ws = ctx.workspace(F)
dev = ctx.device
b = ctx.make_batch(torch.from_numpy(host["disparity_now"]).to(dev)[idx].contiguous(), torch.from_numpy(host["disparity_prev"]).to(dev)[idx].contiguous(),
                   torch.from_numpy(host["flow"]).to(dev)[idx].contiguous(), host["t"][idx], host["q"][idx], host["dt"][idx])
lib = ctx.lib
lib.mod_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
out = (C.c_uint64 * 96)()
for it in range(3):
    ctx.process(b, ws); ctx.synchronize()
    lib.mod_debug_counters(ctx.h, out)
    C.cast(out, C.POINTER(C.c_uint64))  # counters reset by the read
    if it < 2:
        pass
names = ["A load+sync", "runs+sync", "B window tests", "B unions", "sync after B", "C publish", "halo publish", "sync", "D stats"]
tot = sum(out[i] for i in range(9))
for i, n in enumerate(names):
    print(f"{n:16s} {out[i]:14d} cycles  {100.0*out[i]/max(tot,1):5.1f}%   max per wave {out[16+i]:10d}")

print("rows in B", out[13], " (row,dv) iterations", out[9], " pass-2 triggers", out[10], " lanes needing", out[11], " wave_unite calls", out[12], " iterations past the label skip", out[14], " of which with a differing label", out[15])

print("k_median wall-clock sums (100 MHz ticks -> us): load+range %.1f  rounds %.1f  exact rank %.1f  ties %.1f" % tuple(out[i] / 100.0 for i in (26, 27, 28, 29)))

print("k_median blocks: busy %d, mean busy-block life %.1f us, first start -> last end %.1f us" % (out[41], out[40] / max(out[41], 1) / 100.0, (out[43] - out[42]) / 100.0))
