"""Event counts / cycles of k_ccl_rows per active tile (diagnostic build: make -C moving_object_detector_amd/csrc PHASE_COUNTERS=1 OUT=../libmod_sf_pc.so;
run with MOD_SF_LIB=.../libmod_sf_pc.so MOD_DEBUG=128 MOD_TILE_KERNEL=rows).  usage: python tools/dbg_rows_counters.py [sequence|pairs]"""
import ctypes as C, os, sys
os.environ["MOD_DEBUG"] = os.environ.get("MOD_DEBUG", "128")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moving_object_detector_amd import synth
from moving_object_detector_amd.pipeline import Context
W, H, F = 1280, 720, 16
if len(sys.argv) > 1 and sys.argv[1] == "pairs":
    cam, host = synth.make_batch(W, H, F, seed=0)
else:
    cam, s = synth.make_sequence(W, H, F, seed=4)
    host = {"disparity_now": s["disparity"][1:], "disparity_prev": s["disparity"][:-1], "flow": s["flow"], "t": s["t"], "q": s["q"], "dt": s["dt"]}
ctx = Context(W, H, max_frames=F)
ctx.set_camera(cam); ctx.set_params(synth.Params())
ws = ctx.workspace(F)
dev = ctx.device
b = ctx.make_batch(*(torch.from_numpy(np.ascontiguousarray(host[k])).to(dev) for k in ("disparity_now", "disparity_prev", "flow")), host["t"], host["q"], host["dt"])
lib = ctx.lib
lib.mod_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
out = (C.c_uint64 * 64)()
for it in range(2):
    ctx.process(b, ws); ctx.synchronize()
    lib.mod_debug_counters(ctx.h, out)
names = ["active tiles", "rounds", "sweeps", "sweep rows", "seg-min calls", "window rows", "window positions C!=0", "positions gated", "D events", "roots",
         "cycles total", "requests", "cycles setup", "cycles sweeps", "cycles window", "cycles output"]
t = max(int(out[0]), 1)
for i, n in enumerate(names):
    print(f"{n:24s} {int(out[i]):12d}   per active tile {out[i] / t:10.2f}")
