"""The shader clock the chip holds inside k_sgm_paths_all after three seconds of back-to-back estimator calls (MI355X_MICROARCH.md "DVFS
give-back", item 6: delta s_memtime / delta s_memrealtime x 100 MHz, stamped inside the kernel).  Needs a DIAGNOSTIC build that was never
committed: k_sgm_paths_all took one more argument (the context's dbg words) and every 64th workgroup's first thread added its
clock64() and wall_clock64() deltas to dbg[44] / dbg[45] when it left (eight lines in sgm.hip, mod_launch.h, mod_sf.hip; built with
PHASE_COUNTERS=1 for mod_debug_counters).  Round 5: 0.488 ms per 720p frame, 2.157 GHz — against 2.10-2.12 GHz from GRBM_GUI_ACTIVE
in the counter passes and the 2.4 GHz the VALU bound had been priced at.
MOD_SF_LIB=.../libmod_sf_clk.so MOD_DEBUG=0 python tools/sgm_clock_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from moving_object_detector_amd import capi, synth
from moving_object_detector_amd.pipeline import Context
W, H, D, F = 1280, 720, 128, 16
pairs = [synth.make_stereo_images(W, H, 7 + f, D, n_boxes=5) for f in range(F)]
ctx = Context(W, H, max_frames=F); ctx.set_camera(synth.make_camera(W, H)); ctx.set_params(synth.Params())
dev = ctx.device
tl = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev); tr = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
out = torch.empty((F, H, W), dtype=torch.float32, device=dev)
prm = capi.ModSgmParams(D, 6, 96, 8, 1, 1)
lib = ctx.lib; lib.mod_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
cnt = (C.c_uint64 * 96)()
t_end = time.time() + 3.0
n = 0
while time.time() < t_end:                       # three seconds of back-to-back estimator calls: the clock the chip HOLDS
    lib.mod_sgm_compute_dev(ctx.h, F, tl.data_ptr(), tr.data_ptr(), C.byref(prm), out.data_ptr()); n += 1
    if n % 8 == 0: ctx.synchronize()
ctx.synchronize()
lib.mod_debug_counters(ctx.h, cnt)               # (reset)
t0 = time.perf_counter()
for _ in range(10):
    lib.mod_sgm_compute_dev(ctx.h, F, tl.data_ptr(), tr.data_ptr(), C.byref(prm), out.data_ptr())
ctx.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / (10 * F)
lib.mod_debug_counters(ctx.h, cnt)
print(f"{ms:.3f} ms per frame; shader clock inside k_sgm_paths_all (sampled waves, s_memtime / s_memrealtime x 100 MHz): {cnt[44] / max(cnt[45], 1) * 0.1:.3f} GHz")
