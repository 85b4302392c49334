"""Time of the on-GPU disparity estimator (mod_sgm_compute_dev) per frame.  usage (GPU box): python tools/time_sgm.py [W H [D]]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moving_object_detector_amd import capi, synth
from moving_object_detector_amd.pipeline import Context
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1280, 720)
D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
F = int(os.environ.get("SGM_F", "16"))
pairs = [synth.make_stereo_images(W, H, 7 + f, D, n_boxes=5) for f in range(F)]
ctx = Context(W, H, max_frames=F)
ctx.set_camera(synth.make_camera(W, H)); ctx.set_params(synth.Params())
dev = ctx.device
tl = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev); tr = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
out = torch.empty((F, H, W), dtype=torch.float32, device=dev)
for paths in (8, 4):
    prm = capi.ModSgmParams(D, 6, 96, paths, 1, 1)
    for _ in range(2):
        assert ctx.lib.mod_sgm_compute_dev(ctx.h, F, tl.data_ptr(), tr.data_ptr(), C.byref(prm), out.data_ptr()) == 0
    ctx.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        ctx.lib.mod_sgm_compute_dev(ctx.h, F, tl.data_ptr(), tr.data_ptr(), C.byref(prm), out.data_ptr())
    ctx.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / (reps * F)
    print(f"{W}x{H} D={D} paths={paths}: {ms:.2f} ms per frame ({1e3 / ms:.0f} frames/s); valid {float((out >= 0).float().mean()):.2f}")
