"""Soak of the on-GPU disparity estimator: random image sizes (heights / widths that are and are not multiples of 4: the four-line path
kernels' UNIFORM and ragged variants), disparity counts (128 = four-line kernels in one grid, others = one-line kernels), penalties,
4 / 8 paths, frame counts across group boundaries — the complete estimator against the CPU restatement (oracle/sgm_ref.cpp), bit for bit.
usage (GPU box): python tools/soak_sgm.py [seconds] [seed]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from moving_object_detector_amd import capi, synth
from moving_object_detector_amd.pipeline import Context
from oracle import pysgm

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
pysgm.lib()
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    W = int(rng.choice([9, 16, 33, 64, 70, 100, 131, 160, 200, 257, 320]))
    H = int(rng.choice([7, 8, 12, 24, 33, 48, 61, 96, 120]))
    D = int(rng.choice([128, 128, 128, 64, 100, 33, 16, 8]))
    F = int(rng.choice([1, 2, 3, 8, 9, 17]))
    paths = int(rng.choice([8, 8, 4]))
    P1 = int(rng.integers(1, 12)); P2 = int(rng.integers(P1 + 1, 120))
    lr, med = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    seed = int(rng.integers(0, 1 << 30))
    pairs = [synth.make_stereo_images(W, H, seed + f, D, n_boxes=2) for f in range(F)]
    left, right = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    if it % 5 == 0:                                    # flat / saturated images: every cost ties
        left[:] = 200; right[:] = 200
    ctx = Context(W, H, max_frames=F)
    ctx.set_camera(synth.make_camera(W, H)); ctx.set_params(synth.Params())
    dev = ctx.device
    prm = capi.ModSgmParams(D, P1, P2, paths, int(lr), int(med))
    out = torch.full((F, H, W), -7.0, dtype=torch.float32, device=dev)
    tl, tr = torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev)
    rc = ctx.lib.mod_sgm_compute_dev(ctx.h, F, tl.data_ptr(), tr.data_ptr(), C.byref(prm), out.data_ptr())
    assert rc == 0, ctx.lib.mod_last_error(ctx.h)
    ctx.synchronize()
    got = out.cpu().numpy()
    ctx.close()
    for f in sorted({0, F // 2, F - 1}):
        want = pysgm.compute(left[f], right[f], D, P1, P2, paths, lr, med)
        if not np.array_equal(got[f], want):
            print("MISMATCH", dict(W=W, H=H, D=D, F=F, paths=paths, P1=P1, P2=P2, lr=lr, med=med, seed=seed, f=f), int((got[f] != want).sum())); sys.exit(1)
    print(f"ok #{it}: {W}x{H}x{F} D={D} paths={paths} P1={P1} P2={P2} lr={lr} median={med}", flush=True)
print("sgm soak passed:", it, "configurations")
