"""Debug aid for the tile kernels: runs one synthetic case through the HIP path and the oracle and describes where labels differ.
usage (GPU box): python tools/dbg_rows.py [W H seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moving_object_detector_amd import synth
from moving_object_detector_amd.pipeline import Context
from oracle import pyoracle

W, H, seed = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (386, 333, 77)
cam, bq = synth.make_batch(W, H, 3, seed=seed)
bq["flow"] = (np.round(bq["flow"] * 2) / 2).astype(np.float32)
bq["disparity_now"] = np.round(bq["disparity_now"]).astype(np.float32)
prm = synth.Params(dynamic_flow_diff=1, cluster_size=200, dynamic_speed=0.01)
F = 3
ctx = Context(W, H, max_frames=F, max_objects=W * H // 100)
ctx.set_camera(cam); ctx.set_params(prm)
ws = ctx.workspace(F)
dev = ctx.device
b = ctx.make_batch(*(torch.from_numpy(np.ascontiguousarray(bq[k])).to(dev) for k in ("disparity_now", "disparity_prev", "flow")), bq["t"], bq["q"], bq["dt"])
assert ctx.process(b, ws) == 0
ctx.synchronize()
labels = ws["labels"].cpu().numpy()
mask = ws["mask"].cpu().numpy()
for f in range(F):
    ref = pyoracle.construct(cam, prm, bq["disparity_now"][f], bq["disparity_prev"][f], bq["flow"][f], bq["t"][f], bq["q"][f], float(bq["dt"][f]), "tidy")
    lab, ro, K = pyoracle.cluster(ref, prm, "tidy", max_objects=W * H)
    g = labels[f]
    bad = np.argwhere(g != lab)
    print(f"frame {f}: oracle K={K} gpu K={int(ws['n_clusters'][f])} mismatching pixels {len(bad)} of {W*H}; labelled gpu {(g>=0).sum()} oracle {(lab>=0).sum()}")
    if len(bad):
        y, x = bad[0]
        print(" first mismatch at", (y, x), "tile", (y // 16, x // 64), "in-tile", (y % 16, x % 64), "gpu", g[y, x], "oracle", lab[y, x])
        # sizes of clusters
        gs = np.bincount(g[g >= 0], minlength=1); os_ = np.bincount(lab[lab >= 0], minlength=1)
        print(" gpu sizes", gs[:12].tolist(), "oracle sizes", os_[:12].tolist())
        # is the partition the same up to renumbering?
        pairs = set(zip(g[(g >= 0) | (lab >= 0)].tolist(), lab[(g >= 0) | (lab >= 0)].tolist()))
        print(" distinct (gpu, oracle) label pairs:", sorted(pairs)[:24], len(pairs))
        y0, y1, x0, x1 = max(0, y - 6), min(H, y + 7), max(0, x - 8), min(W, x + 9)
        print(" gpu labels:\n", g[y0:y1, x0:x1]); print(" oracle labels:\n", lab[y0:y1, x0:x1])
ctx.close()
