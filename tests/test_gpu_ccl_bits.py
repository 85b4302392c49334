"""GPU: the bit-plane tile kernel (k_ccl_bits<n>, one instance per neighbor_distance 1 .. 10) on the cases its construction distinguishes — through the
clusterer alone (mod_cluster_dev on caller-supplied planes) against the oracle, labels and objects bit for bit:
  * ONE depth class (constant depth): every tile stays on the bit path; one-component-by-inspection tiles, flooded tiles (several
    components, holes, gaps around the window size), singletons, image borders, widths that are not a multiple of 64;
  * several classes: depth levels more than depth_diff apart (two / three / five of them: five exceed the kernel's four classes and
    send the tile to the union-find kernel), levels LESS than depth_diff apart that chain (one class must not be split), a ramp
    (a class that is not clear of the next: union-find kernel), NaN depth on dynamic pixels (links with everything: union-find kernel);
  * a batch whose frames differ (per-frame lists and counters of the two tile kernels).
Semantics: clusterer_nodelet.cpp:56-83,186-219."""
import numpy as np
import pytest

from test_gpu_cluster_stress import _check, _make_cloud

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _prm(csize=3, n=4):
    from moving_object_detector_amd import synth
    return synth.Params(cluster_size=csize, neighbor_distance=n, depth_diff=0.15, dynamic_speed=0.3)


@pytest.mark.parametrize("shape", [(200, 150), (320, 16), (64, 200), (131, 77), (640, 480), (40, 30), (17, 20), (65, 17)])
@pytest.mark.parametrize("density", [0.01, 0.08, 0.3, 0.97])
def test_one_class_noise(oracle, shape, density):
    W, H = shape
    rng = np.random.default_rng(W * 7 + int(density * 100))
    _check(oracle, _make_cloud(W, H, rng.random((H, W)) < density, np.full((H, W), 5.0), rng), _prm(), W, H)


@pytest.mark.parametrize("pattern", ["vbars5", "vbars6", "hbars5", "hbars6", "blobs", "blob_with_holes", "frame", "stairs"])
def test_one_class_patterns(oracle, pattern):
    W, H = 260, 150
    rng = np.random.default_rng(9)
    ys, xs = np.mgrid[0:H, 0:W]
    d = np.zeros((H, W), bool)
    if pattern.startswith("vbars"):
        d[:, ::int(pattern[-1])] = True                           # gap 5: linked (distance <= 4 + 1); gap 6: every bar alone
    elif pattern.startswith("hbars"):
        d[::int(pattern[-1]), :] = True
    elif pattern == "blobs":
        for (y0, y1, x0, x1) in ((10, 60, 30, 150), (20, 28, 170, 200), (22, 23, 210, 211), (40, 44, 5, 20), (70, 140, 60, 70), (100, 104, 80, 250)):
            d[y0:y1, x0:x1] = True
    elif pattern == "blob_with_holes":
        d[5:140, 5:250] = rng.random((135, 245)) < 0.94            # 6 % holes like invalid disparities; a rare gap of >= 4 splits a row's run
        d[60:70, 100:140] = False
    elif pattern == "frame":
        d[8:120, 8:240] = True; d[14:114, 14:234] = False          # a ring: the flood has to go round
    else:
        for k in range(20):                                         # a staircase going UP to the right: rows link only through the window
            d[140 - 6 * k:143 - 6 * k, 10 + 11 * k:24 + 11 * k] = True
    _check(oracle, _make_cloud(W, H, d, np.full((H, W), 4.0), rng), _prm(2), W, H)


@pytest.mark.parametrize("levels,step", [(2, 1.0), (3, 0.4), (5, 0.3), (3, 0.1), (8, 0.05)])
def test_depth_classes(oracle, levels, step):
    """levels x step: 2 x 1.0 / 3 x 0.4 / 5 x 0.3 are classes clear of each other (5 of them: more than the kernel peels off); 3 x 0.1
    are adjacent levels that link pairwise (0.1 <= 0.15) but not end to end (0.2 > 0.15): not classes at all; 8 x 0.05 is a ramp."""
    W, H = 200, 150
    rng = np.random.default_rng(levels * 10 + int(step * 100))
    for layout in ("columns", "random", "halves"):
        if layout == "columns":
            z = 5.0 + step * ((np.arange(W)[None, :] // 7) % levels) + np.zeros((H, 1))
        elif layout == "random":
            z = 5.0 + step * rng.integers(0, levels, size=(H, W))
        else:
            z = 5.0 + step * (np.arange(H)[:, None] * levels // H) + np.zeros((1, W))
        for density in (0.25, 1.0):
            _check(oracle, _make_cloud(W, H, rng.random((H, W)) < density, z, rng), _prm(), W, H)


def test_nan_and_infinite_depth_on_dynamic_pixels(oracle):
    W, H = 200, 150
    rng = np.random.default_rng(4)
    z = np.full((H, W), 6.0)
    z[:, 100:] = 7.0
    z[rng.random((H, W)) < 0.02] = np.nan                           # NaN links with everything: bridges the two depths
    _check(oracle, _make_cloud(W, H, rng.random((H, W)) < 0.5, z, rng), _prm(), W, H)
    z = np.full((H, W), 6.0)
    z[40:60, :] = np.inf                                            # inf - inf = NaN links, inf - 6 does not: a class of its own
    z[100:110, :] = -np.inf
    _check(oracle, _make_cloud(W, H, np.ones((H, W), bool), z, rng), _prm(), W, H)


def test_batch_with_mixed_tiles(oracle):
    from moving_object_detector_amd import synth
    from moving_object_detector_amd.pipeline import Context, PLANES
    from util import compare_objects
    W, H, F = 192, 96, 6
    rng = np.random.default_rng(12)
    prm = _prm(4)
    zs = [np.full((H, W), 5.0), 5.0 + 0.5 * rng.integers(0, 2, size=(H, W)), 5.0 + 0.05 * rng.integers(0, 9, size=(H, W)),
          5.0 + 0.3 * rng.integers(0, 6, size=(H, W)), np.full((H, W), 9.0), 5.0 + 1.0 * (np.arange(W)[None, :] > 90) + np.zeros((H, 1))]
    clouds = [_make_cloud(W, H, rng.random((H, W)) < d, z, rng) for d, z in zip((0.6, 0.5, 0.7, 0.4, 0.0, 1.0), zs)]
    ctx = Context(W, H, max_frames=F, max_objects=W * H // prm.cluster_size + 1)
    ctx.set_camera(synth.make_camera(W, H)); ctx.set_params(prm)
    ws = ctx.workspace(F)
    for f, cl in enumerate(clouds):
        for i, k in enumerate(PLANES):
            ws["planes"][i, f].copy_(torch.from_numpy(cl[k]))
    for _ in range(2):                                               # the second call runs over the lists / headers of the first
        assert ctx.cluster(F, ws, mask_ready=False) == 0
    ctx.synchronize()
    labels, objs = ws["labels"].cpu().numpy(), ctx.objects_to_host(ws)
    for f, cl in enumerate(clouds):
        rl, ro, _ = oracle.cluster(cl, prm, "tidy", max_objects=W * H)
        assert np.array_equal(labels[f], rl), f
        compare_objects(objs[f], ro, strict_velocity=True)
    ctx.close()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 6, 7, 8, 9, 10])
def test_other_window_sizes(oracle, n):
    """Every instance of the kernel (Clusterer.cfg:11 allows 1 .. 10): one class (noise at three densities, bars with gaps of n and
    n + 1 cells: linked / not linked), two classes side by side, a fragmented tile (more pieces than the kernel floods one by one:
    the union-find kernel takes over), widths that are not a multiple of 64."""
    W, H = 203, 100
    rng = np.random.default_rng(100 + n)
    flat = np.full((H, W), 5.0)
    for density in (0.03, 0.2, 0.9):
        _check(oracle, _make_cloud(W, H, rng.random((H, W)) < density, flat, rng), _prm(2, n), W, H)
    for gap in (n, n + 1, n + 2):
        d = np.zeros((H, W), bool); d[:, ::gap] = True
        _check(oracle, _make_cloud(W, H, d, flat, rng), _prm(2, n), W, H)
        d = np.zeros((H, W), bool); d[::gap, :] = True
        _check(oracle, _make_cloud(W, H, d, flat, rng), _prm(2, n), W, H)
    z = flat.copy(); z[:, 90:] = 5.5
    _check(oracle, _make_cloud(W, H, rng.random((H, W)) < 0.6, z, rng), _prm(2, n), W, H)
    ys, xs = np.mgrid[0:H, 0:W]
    d = ((xs % (2 * n + 3)) < 2) & ((ys % (n + 2)) < 1)              # dozens of two-pixel pieces per tile
    _check(oracle, _make_cloud(W, H, d, flat, rng), _prm(2, n), W, H)
