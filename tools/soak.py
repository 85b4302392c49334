"""Soak: random cameras / parameters / sizes / seeds through the HIP path against the CPU oracle, bit for bit.
usage (GPU box): python tools/soak.py [seconds] [seed]   — prints a line per configuration, exits 1 on the first mismatch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from moving_object_detector_amd import synth
from moving_object_detector_amd.pipeline import Context
from moving_object_detector_amd import capi
from oracle import pyoracle
from util import PLANES, bits_equal, compare_objects, first_mismatch

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
pyoracle.lib()
# MOD_SF_LIB=.../libmod_sf_checked.so: the CHECKED build (csrc/mod_device.h MOD_CHECK) counts every data-derived index that fails its
# test instead of using it; the soak then also requires all those counters to stay 0
CHECKED = "checked" in os.path.basename(capi.LIB_PATH)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    W = int(rng.choice([64, 100, 131, 160, 250, 257, 320, 386, 500, 640]))
    H = int(rng.choice([17, 48, 67, 96, 120, 240, 333]))
    F = int(rng.choice([1, 2, 3, 5]))
    if it % 50 == 0:                                   # now and then a full-size frame (BASELINE.json's sizes)
        W, H, F = [(1280, 720, 2), (1920, 1080, 1), (1242, 376, 3)][(it // 50) % 3]
    seed = int(rng.integers(0, 1 << 30))
    cam, batch = synth.make_batch(W, H, F, seed=seed)
    prm = synth.Params(dynamic_flow_diff=int(rng.choice([1, 2, 5, 8])), cluster_size=int(rng.choice([1, 5, 40, 200, 2500])),
                       neighbor_distance=int(rng.integers(1, 11)), depth_diff=float(rng.choice([0.01, 0.05, 0.15, 1.0])),
                       dynamic_speed=float(rng.choice([0.01, 0.1, 0.3, 1.0])))
    mode = int(rng.integers(0, 3))
    if mode == 1:       # quantised flow / disparity: many equal norms -> tied medians
        batch["flow"] = (np.round(batch["flow"] * 2) / 2).astype(np.float32)
        batch["disparity_now"] = (np.round(batch["disparity_now"]) ).astype(np.float32)
    if mode == 2:       # sprinkle hostile values
        m = rng.random(batch["disparity_now"].shape) < 0.02
        batch["disparity_now"][m] = rng.choice(np.array([np.nan, np.inf, 0.0, -1.0, 1e-30, 500.0], np.float32), int(m.sum()))
        m = rng.random(batch["flow"].shape) < 0.02
        batch["flow"][m] = rng.choice(np.array([np.nan, 1e9, -1e9, 0.5, 2147483648.0], np.float32), int(m.sum()))
    ctx = Context(W, H, max_frames=F, max_objects=W * H // max(prm.cluster_size, 1) + 1)
    ctx.set_camera(cam); ctx.set_params(prm)
    ws = ctx.workspace(F, extras=True)
    dev = ctx.device
    b = ctx.make_batch(torch.from_numpy(batch["disparity_now"]).to(dev), torch.from_numpy(batch["disparity_prev"]).to(dev),
                       torch.from_numpy(batch["flow"]).to(dev), batch["t"], batch["q"], batch["dt"])
    assert ctx.process(b, ws) == 0
    ctx.synchronize()
    planes = ws["planes"].cpu().numpy(); labels = ws["labels"].cpu().numpy(); objs = ctx.objects_to_host(ws)
    sflow = ws["static_flow"].cpu().numpy(); nclu = ws["n_clusters"].cpu().numpy()
    if CHECKED:
        import ctypes as C
        cnt = (C.c_uint64 * 96)()
        ctx.lib.mod_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        assert ctx.lib.mod_debug_counters(ctx.h, cnt) == 0
        bad = {i - 64: int(cnt[i]) for i in range(64, 96) if cnt[i]}
        if bad:
            print("MISMATCH index check (code: count)", bad, dict(W=W, H=H, F=F, seed=seed, prm=prm)); sys.exit(1)
    ctx.close()
    nobj = namb = 0
    for f in range(F):
        ref = pyoracle.construct(cam, prm, batch["disparity_now"][f], batch["disparity_prev"][f], batch["flow"][f], batch["t"][f],
                                 batch["q"][f], float(batch["dt"][f]), "tidy")
        for i, k in enumerate(PLANES):
            if not bits_equal(planes[i, f], ref[k]):
                print("MISMATCH plane", k, dict(W=W, H=H, F=F, seed=seed, f=f, prm=prm), first_mismatch(planes[i, f], ref[k])); sys.exit(1)
        if not bits_equal(sflow[f], ref["static_flow"]):
            print("MISMATCH static flow", dict(W=W, H=H, seed=seed, f=f)); sys.exit(1)
        rl, ro, K = pyoracle.cluster(ref, prm, "tidy", max_objects=W * H)
        if not np.array_equal(labels[f], rl) or int(nclu[f]) != K:
            print("MISMATCH labels", dict(W=W, H=H, F=F, seed=seed, f=f, prm=prm), int((labels[f] != rl).sum())); sys.exit(1)
        try:
            compare_objects(objs[f], ro, strict_velocity=True)
        except AssertionError as e:
            print("MISMATCH objects", dict(W=W, H=H, F=F, seed=seed, f=f, prm=prm), e); sys.exit(1)
        nobj += len(ro)
    print(f"ok #{it}: {W}x{H}x{F} seed {seed} mode {mode} n={prm.neighbor_distance} cs={prm.cluster_size} dd={prm.depth_diff} ds={prm.dynamic_speed} fd={prm.dynamic_flow_diff}: {nobj} objects", flush=True)
print("soak passed:", it, "configurations" + (" (checked build, no index violation)" if CHECKED else ""))
