import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from moving_object_detector_amd import synth
from oracle import pyoracle
import test_gpu_cluster_stress as T
density, n = float(sys.argv[1]), int(sys.argv[2])
W, H = 200, 150
rng = np.random.default_rng(int(density * 100) * 31 + n)
dyn = rng.random((H, W)) < density
z = 5.0 + 0.1 * rng.integers(0, 4, size=(H, W))
prm = synth.Params(cluster_size=3, neighbor_distance=n, depth_diff=0.15, dynamic_speed=0.3)
planes = T._make_cloud(W, H, dyn, z, rng)
lab, objs, K = T._cluster_gpu(planes, prm, W, H)
rl, ro, rK = pyoracle.cluster(planes, prm, "tidy", max_objects=W*H)
print("K", K, rK)
# components: map oracle label -> set of gpu labels
bad = []
for k in range(rK):
    g = np.unique(lab[rl == k])
    if g.size != 1: bad.append((k, g, int((rl==k).sum())))
print("oracle clusters split on GPU:", bad[:5])
for k, g, sz in bad[:1]:
    ys, xs = np.nonzero(rl == k)
    print("bbox y", ys.min(), ys.max(), "x", xs.min(), xs.max(), "size", sz)
    for gl in g:
        yy, xx = np.nonzero((rl == k) & (lab == gl)); print(" gpu label", gl, "count", yy.size, "y", yy.min(), yy.max(), "x", xx.min(), xx.max())
# also GPU clusters merged wrongly
for k in range(K):
    o = np.unique(rl[lab == k])
    if o.size != 1: print("gpu cluster", k, "spans oracle labels", o[:5]); break
