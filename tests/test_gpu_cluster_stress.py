"""GPU: the clusterer alone (mod_cluster_dev on caller-supplied planes) against the oracle on adversarial clouds —
random dynamic masks of several densities, few / many depth levels, diagonal and checkerboard patterns, every
neighbor_distance class (NMAX 4 / 8 / 16 kernels), tiles with more than 32 components (the LDS-slot overflow path),
NaN depth on dynamic pixels (NaN links, clusterer_nodelet.cpp:194)."""
import ctypes as C

import numpy as np
import pytest

from util import compare_objects

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _cluster_gpu(planes, prm, W, H):
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.pipeline import Context, PLANES
    ctx = Context(W, H, max_frames=1, max_objects=W * H // prm.cluster_size + 1)
    ctx.set_camera(synth.make_camera(W, H))
    ctx.set_params(prm)
    ws = ctx.workspace(1)
    for i, k in enumerate(PLANES):
        ws["planes"][i, 0].copy_(torch.from_numpy(planes[k]))
    assert ctx.cluster(1, ws, mask_ready=False) == 0
    ctx.synchronize()
    out = ws["labels"][0].cpu().numpy(), ctx.objects_to_host(ws)[0], int(ws["n_clusters"][0])
    # the same cloud without a labels plane (the reference renders its cluster image only for subscribers): identical objects
    raw, n_obj = ws["objects"][0].cpu().numpy().copy(), int(ws["n_objects"][0])
    ws2 = ctx.workspace(1, labels=False)
    assert ws2["labels"] is None
    for i, k in enumerate(PLANES):
        ws2["planes"][i, 0].copy_(torch.from_numpy(planes[k]))
    ws2["objects"].zero_()
    assert ctx.cluster(1, ws2, mask_ready=False) == 0
    ctx.synchronize()
    assert int(ws2["n_objects"][0]) == n_obj and int(ws2["n_clusters"][0]) == out[2]
    assert np.array_equal(ws2["objects"][0].cpu().numpy()[:n_obj], raw[:n_obj]), "objects differ without the labels plane"
    ctx.close()
    return out


def _make_cloud(W, H, dyn, z, rng):
    """Planes with the given dynamic mask and depth; velocities are distinct per pixel so that medians are unambiguous."""
    vx = np.where(dyn, 1.0 + rng.random((H, W)), 0.0).astype(np.float32)
    vy = np.where(dyn, 0.1 * rng.standard_normal((H, W)), 0.0).astype(np.float32)
    vz = np.where(dyn, 0.1 * rng.standard_normal((H, W)), 0.0).astype(np.float32)
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    return {"x": (xs * 0.01).astype(np.float32), "y": (ys * 0.01).astype(np.float32), "z": z.astype(np.float32), "vx": vx, "vy": vy, "vz": vz}


def _check(oracle, planes, prm, W, H):
    lab, objs, K = _cluster_gpu(planes, prm, W, H)
    rl, ro, rK = oracle.cluster(planes, prm, "tidy", max_objects=W * H // 2 + 16)
    assert K == rK, (K, rK)
    assert np.array_equal(lab, rl), int((lab != rl).sum())
    compare_objects(objs, ro, strict_velocity=True)


@pytest.mark.parametrize("density", [0.02, 0.2, 0.5, 0.9])
@pytest.mark.parametrize("n", [1, 4, 7, 16])
def test_random_masks(oracle, density, n):
    from moving_object_detector_amd import synth
    W, H = 200, 150
    rng = np.random.default_rng(int(density * 100) * 31 + n)
    dyn = rng.random((H, W)) < density
    z = 5.0 + 0.1 * rng.integers(0, 4, size=(H, W))          # 4 depth levels 0.1 m apart: gate 0.15 m splits non-adjacent ones
    prm = synth.Params(cluster_size=3, neighbor_distance=n, depth_diff=0.15, dynamic_speed=0.3)
    _check(oracle, _make_cloud(W, H, dyn, z, rng), prm, W, H)


@pytest.mark.parametrize("pattern", ["anti_diagonals", "checkerboard", "rows", "columns", "dots"])
def test_patterns(oracle, pattern):
    from moving_object_detector_amd import synth
    W, H = 192, 96
    rng = np.random.default_rng(5)
    ys, xs = np.mgrid[0:H, 0:W]
    dyn = {"anti_diagonals": (xs + ys) % 3 == 0,           # up-right neighbours are never tested: every pixel stays alone at n=1
           "checkerboard": (xs + ys) % 2 == 0,
           "rows": ys % 3 == 0,
           "columns": xs % 3 == 0,
           "dots": (xs % 5 == 0) & (ys % 5 == 0)}[pattern]
    z = np.full((H, W), 4.0)
    for n in (1, 2, 4):
        prm = synth.Params(cluster_size=2, neighbor_distance=n)
        _check(oracle, _make_cloud(W, H, dyn, z, rng), prm, W, H)


def test_many_components_per_tile_and_nan_depth(oracle):
    """More than 32 components inside one 64x16 tile (overflow of the LDS statistics slots) and NaN depth on dynamic
    pixels, which the reference links to everything in the window (NaN > th is false)."""
    from moving_object_detector_amd import synth
    W, H = 256, 64
    rng = np.random.default_rng(11)
    ys, xs = np.mgrid[0:H, 0:W]
    dyn = ((xs % 4) < 2) & ((ys % 4) < 2)                   # 2x2 blobs on a 4-px grid: 16 x 4 = 64 components per tile at n=1
    z = np.full((H, W), 6.0)
    planes = _make_cloud(W, H, dyn, z, rng)
    _check(oracle, planes, synth.Params(cluster_size=2, neighbor_distance=1), W, H)
    _check(oracle, planes, synth.Params(cluster_size=2, neighbor_distance=2), W, H)   # now everything is one component
    z2 = 5.0 + 1.0 * rng.integers(0, 3, size=(H, W)).astype(np.float64)
    z2[rng.random((H, W)) < 0.05] = np.nan
    planes = _make_cloud(W, H, rng.random((H, W)) < 0.6, z2, rng)
    _check(oracle, planes, synth.Params(cluster_size=4, neighbor_distance=3), W, H)


def test_batch_of_frames_with_different_content(oracle):
    """Frames of one batch do not leak into each other (counters, records, lists are per frame)."""
    from moving_object_detector_amd import synth
    from moving_object_detector_amd.pipeline import Context, PLANES
    W, H, F = 160, 96, 5
    rng = np.random.default_rng(3)
    prm = synth.Params(cluster_size=5, neighbor_distance=3)
    clouds = [_make_cloud(W, H, rng.random((H, W)) < d, 5.0 + 0.1 * rng.integers(0, 3, size=(H, W)), rng) for d in (0.0, 0.7, 0.05, 0.4, 1.0)]
    ctx = Context(W, H, max_frames=F, max_objects=W * H // prm.cluster_size + 1)
    ctx.set_camera(synth.make_camera(W, H))
    ctx.set_params(prm)
    ws = ctx.workspace(F)
    for f, cl in enumerate(clouds):
        for i, k in enumerate(PLANES):
            ws["planes"][i, f].copy_(torch.from_numpy(cl[k]))
    for _ in range(2):                                       # second call re-uses every scratch buffer
        assert ctx.cluster(F, ws, mask_ready=False) == 0
    ctx.synchronize()
    labels = ws["labels"].cpu().numpy()
    objs = ctx.objects_to_host(ws)
    for f, cl in enumerate(clouds):
        rl, ro, rK = oracle.cluster(cl, prm, "tidy", max_objects=W * H)
        assert np.array_equal(labels[f], rl), f
        compare_objects(objs[f], ro, strict_velocity=True)
    ctx.close()
