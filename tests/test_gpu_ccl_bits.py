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


def test_image_narrower_than_the_window(oracle):
    """W < neighbor_distance (allowed: max_width >= 1, n up to 10): the halo-column depth loads of tile column 0 stay inside the
    image's rows (round 4 read up to n - 1 floats past the end of the plane's last row)."""
    rng = np.random.default_rng(77)
    for (W, H, n) in ((8, 8, 10), (3, 40, 7), (1, 30, 4), (9, 17, 10)):
        _check(oracle, _make_cloud(W, H, rng.random((H, W)) < 0.7, np.full((H, W), 5.0), rng), _prm(2, n), W, H)


def _fused(oracle, cam, prm, z, dyn, shift=6.0):
    """One frame through the FUSED path (mod_process_dev: the scene-flow epilogue writes mask words + their depth ranges, k_ccl_bits
    decides one-class tiles from the ranges) against the oracle.  A camera at rest, previous disparity = now, flow = `shift` px on the
    pixels that are to be dynamic and 0 elsewhere: with dynamic_flow_diff = 1 exactly those (where the warp stays in the image) move."""
    from moving_object_detector_amd.pipeline import Context, PLANES
    from util import bits_equal, compare_objects
    H, W = z.shape
    fT = np.float32(cam.disp_f) * np.float32(cam.disp_T)
    d = (fT / z.astype(np.float32)).astype(np.float32)
    flow = np.zeros((H, W, 2), np.float32)
    flow[..., 0] = np.where(dyn, np.float32(shift), np.float32(0.0))
    t, q, dt = np.zeros(3), np.array([0.0, 0.0, 0.0, 1.0]), 0.1
    ctx = Context(W, H, max_frames=1, max_objects=W * H // prm.cluster_size + 1)
    ctx.set_camera(cam); ctx.set_params(prm)
    ws = ctx.workspace(1)
    dev = ctx.device
    dn = torch.from_numpy(d[None]).to(dev)
    b = ctx.make_batch(dn, dn.clone(), torch.from_numpy(flow[None]).to(dev), t[None], q[None], [dt])
    for _ in range(2):                                               # the second call runs over the first one's ranges / headers
        assert ctx.process(b, ws) == 0
    ctx.synchronize()
    ref = oracle.construct(cam, prm, d, d, flow, t, q, dt, "tidy")
    planes = ws["planes"].cpu().numpy()
    for i, k in enumerate(PLANES):
        assert bits_equal(planes[i, 0], ref[k]), k
    rl, ro, K = oracle.cluster(ref, prm, "tidy", max_objects=W * H)
    assert int(ws["n_clusters"][0]) == K
    assert np.array_equal(ws["labels"][0].cpu().numpy(), rl), int((ws["labels"][0].cpu().numpy() != rl).sum())
    compare_objects(ctx.objects_to_host(ws)[0], ro, strict_velocity=True)
    ctx.close()
    return int((rl >= 0).sum())


@pytest.mark.parametrize("W,H", [(330, 70), (640, 96), (257, 33), (1242, 40), (131, 50)])
def test_fused_path_depth_ranges(oracle, W, H):
    """The per-word depth ranges of the fused path (round 5) on layouts where they decide: one depth everywhere (every tile takes the
    range path), two depths within depth_diff (still one class), a depth step INSIDE the left neighbour word but outside the halo
    columns (the range test fails although the grid is one class: the row path must give the same answer), steps along word borders,
    across rows, single far pixels inside a near word (the range must cover them), random levels, and widths served by each of the
    three scene-flow kernels (multiple of 4, even, odd)."""
    from moving_object_detector_amd import synth
    cam = synth.make_camera(W, H)
    prm = synth.Params(dynamic_flow_diff=1, cluster_size=3, neighbor_distance=4, depth_diff=0.15, dynamic_speed=0.05)
    rng = np.random.default_rng(W + H)
    xs = np.arange(W)[None, :] + np.zeros((H, 1), int)
    ys = np.arange(H)[:, None] + np.zeros((1, W), int)
    layouts = {
        "flat": np.full((H, W), 5.0),
        "two_near": 5.0 + 0.1 * ((xs // 9 + ys // 5) % 2),
        "step_in_left_word": np.where(xs % 64 < 40, 5.0, 6.0) + 0.0 * ys,      # the step sits 24 columns left of every tile
        "step_at_word_border": np.where((xs // 64) % 2 == 0, 5.0, 6.0) + 0.0 * ys,
        "step_at_halo": np.where((xs + 3) % 64 < 32, 5.0, 6.0) + 0.0 * ys,      # inside the 4 halo columns
        "rows": np.where((ys // 7) % 2 == 0, 5.0, 5.5) + 0.0 * xs,
        "levels": 5.0 + 0.4 * rng.integers(0, 3, size=(H, W)),
    }
    far = np.full((H, W), 5.0)
    far[rng.random((H, W)) < 0.004] = 9.0                                       # lone far pixels inside near words
    layouts["lone_far_pixels"] = far
    total = 0
    for name, z in layouts.items():
        for density in (0.35, 1.0):
            dyn = rng.random((H, W)) < density
            total += _fused(oracle, cam, prm, z, dyn)
    assert total > 0                                                            # the layouts do produce clusters


def test_fused_path_other_window_sizes(oracle):
    from moving_object_detector_amd import synth
    W, H = 200, 60
    cam = synth.make_camera(W, H)
    rng = np.random.default_rng(5)
    xs = np.arange(W)[None, :] + np.zeros((H, 1), int)
    for n in (1, 2, 7, 10, 12):                                                 # 12: outside the bit kernel's range (union-find kernel, no ranges read)
        prm = synth.Params(dynamic_flow_diff=1, cluster_size=2, neighbor_distance=n, depth_diff=0.15, dynamic_speed=0.05)
        for z in (np.full((H, W), 4.0), np.where(xs % 64 < 50, 4.0, 4.6) + 0.0):
            _fused(oracle, cam, prm, z, rng.random((H, W)) < 0.5)
