"""Debug helper for the bit-plane tile kernel: constant-depth clouds (every tile passes k_ccl_bits' depth test) with structured
masks; compares GPU labels with the oracle and reports where they differ.  python tools/dbg_bits.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from moving_object_detector_amd import synth
from oracle import pyoracle
import test_gpu_cluster_stress as T

def case(name, W, H, dyn, csize=3, z=None):
    rng = np.random.default_rng(7)
    z = np.full((H, W), 5.0) if z is None else z
    prm = synth.Params(cluster_size=csize, neighbor_distance=4, depth_diff=0.15, dynamic_speed=0.3)
    planes = T._make_cloud(W, H, dyn, z, rng)
    lab, objs, K = T._cluster_gpu(planes, prm, W, H)
    rl, ro, rK = pyoracle.cluster(planes, prm, "tidy", max_objects=W * H)
    ok = np.array_equal(lab, rl)
    print(f"{name}: K gpu {K} oracle {rK} labels {'OK' if ok else 'MISMATCH'}")
    if ok: return
    bad = np.argwhere(lab != rl)
    print("  mismatching pixels", len(bad), "first", bad[:5].tolist(), "gpu", [int(lab[tuple(b)]) for b in bad[:5]], "oracle", [int(rl[tuple(b)]) for b in bad[:5]])
    tiles = sorted({(int(y) // 16, int(x) // 64) for y, x in bad})
    print("  tiles (ty, wi)", tiles[:20])
    for k in range(rK):
        g = np.unique(lab[rl == k])
        if g.size != 1:
            print("  oracle cluster", k, "size", int((rl == k).sum()), "-> gpu labels", g[:8]); break
    for k in range(K):
        o = np.unique(rl[lab == k])
        if o.size != 1:
            print("  gpu cluster", k, "spans oracle labels", o[:8]); break

W, H = 200, 150
d = np.zeros((H, W), bool); d[20:28, 70:100] = True
case("one blob in one tile", W, H, d)
d = np.zeros((H, W), bool); d[10:60, 30:150] = True
case("blob over many tiles", W, H, d)
d = np.zeros((H, W), bool); d[20:28, 70:100] = True; d[22, 110] = True; d[40:44, 5:20] = True
case("blob + single + blob", W, H, d)
d = np.zeros((H, W), bool); d[33:40, 66:90] = True; d[20:30, 60:64] = True
case("halo only neighbours", W, H, d)
rng = np.random.default_rng(3)
for dens in (0.02, 0.1, 0.3, 0.7):
    case(f"noise {dens}", W, H, rng.random((H, W)) < dens)
d = np.zeros((H, W), bool); d[:, ::5] = True
case("vertical bars gap 5", W, H, d)
d = np.zeros((H, W), bool); d[::5, :] = True
case("horizontal bars gap 5", W, H, d)
d = np.zeros((H, W), bool); d[::6, :] = True
case("horizontal bars gap 6", W, H, d)
d = np.ones((H, W), bool); zz = np.full((H, W), 5.0); zz[:, 90:] = 6.0
case("two depths side by side", W, H, d, z=zz)
zz = np.full((H, W), 5.0); zz[70:, :] = 6.0
case("two depths stacked", W, H, d, z=zz)
zz = 5.0 + 0.1 * (np.arange(W)[None, :] // 7 % 3) + 0.0 * np.arange(H)[:, None]
case("depth steps of 0.1 every 7 columns", W, H, d, z=zz)
W2, H2 = 1280, 720
d = np.zeros((H2, W2), bool); d[100:400, 200:900] = True; zz = np.full((H2, W2), 5.0); zz[:, 600:] = 5.5
case("720p two depths", W2, H2, d, z=zz, csize=2500)

# fused path on synthetic frames (real depths: some tiles pass the bit kernel's depth test, the others go to the union-find kernel)
from moving_object_detector_amd.pipeline import Context
from util import compare_objects
W, H, F = 1280, 720, 3
cam, b = synth.make_batch(W, H, F, seed=0)
prm = synth.Params()
ctx = Context(W, H, max_frames=F)
ctx.set_camera(cam); ctx.set_params(prm)
ws = ctx.workspace(F)
batch = ctx.make_batch(*(torch.from_numpy(np.ascontiguousarray(b[k])).to(ctx.device) for k in ("disparity_now", "disparity_prev", "flow")), b["t"], b["q"], b["dt"])
for rep in range(2):
    assert ctx.process(batch, ws) == 0
    ctx.synchronize()
labels, objs = ws["labels"].cpu().numpy(), ctx.objects_to_host(ws)
th = np.float32(prm.depth_diff)
for f in range(F):
    ref = pyoracle.construct(cam, prm, b["disparity_now"][f], b["disparity_prev"][f], b["flow"][f], b["t"][f], b["q"][f], float(b["dt"][f]), "tidy")
    lab, ro, K = pyoracle.cluster(ref, prm, "tidy", max_objects=W * H)
    ok = np.array_equal(labels[f], lab)
    print(f"fused frame {f}: K oracle {K} labels {'OK' if ok else 'MISMATCH'}")
    try:
        compare_objects(objs[f], ro, strict_velocity=True); print("  objects OK")
    except AssertionError as e:
        print("  objects differ", str(e)[:200])
    if ok: continue
    bad = np.argwhere(labels[f] != lab)
    print("  mismatching pixels", len(bad), "first", bad[:5].tolist(), "gpu", [int(labels[f][tuple(x)]) for x in bad[:5]], "oracle", [int(lab[tuple(x)]) for x in bad[:5]])
    v = np.sqrt(ref["vx"] ** 2 + (ref["vy"] ** 2 + ref["vz"] ** 2)); dyn = v >= np.float32(prm.dynamic_speed)
    for k in range(K):
        g = np.unique(labels[f][lab == k])
        if g.size != 1:
            print("  oracle cluster", k, "size", int((lab == k).sum()), "-> gpu labels", g[:8], [int(((lab == k) & (labels[f] == x)).sum()) for x in g[:8]])
    for k in range(int(labels[f].max()) + 1):
        o = np.unique(lab[labels[f] == k])
        if o.size != 1: print("  gpu cluster", k, "spans oracle labels", o[:8])
