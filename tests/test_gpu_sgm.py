"""GPU: the on-GPU disparity estimator (census transform, the eight aggregation paths with the matching cost computed on the fly,
winner-take-all, median, left-right check) through the C ABI, stage by stage and end to end, bit for bit against the CPU
restatement and the committed fixture; BASELINE config 5's data path at 1920 x 1080."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "sgm_320x240.npz")


def _ctx(W, H, F):
    from moving_object_detector_amd import synth
    from moving_object_detector_amd.pipeline import Context
    ctx = Context(W, H, max_frames=F)
    ctx.set_camera(synth.make_camera(W, H))
    ctx.set_params(synth.Params())
    return ctx


def _stages(ctx, left, right, D, P1=6, P2=96):
    from moving_object_detector_amd import capi
    F, H, W = left.shape
    dev = ctx.device
    dl, dr = torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev)
    cl = torch.empty((F, H, W), dtype=torch.int32, device=dev)
    cr = torch.empty_like(cl)
    assert ctx.lib.mod_sgm_census_dev(ctx.h, F, dl.data_ptr(), cl.data_ptr()) == 0
    assert ctx.lib.mod_sgm_census_dev(ctx.h, F, dr.data_ptr(), cr.data_ptr()) == 0
    prm = capi.ModSgmParams(D, P1, P2, 8, 1, 1)
    out = {}
    for direction in range(8):
        L = torch.full((F, H, W, D), 255, dtype=torch.uint8, device=dev)
        Cv = torch.full((F, H, W, D), 255, dtype=torch.uint8, device=dev)
        assert ctx.lib.mod_sgm_path_dev(ctx.h, F, cl.data_ptr(), cr.data_ptr(), C.byref(prm), direction, L.data_ptr(), Cv.data_ptr()) == 0
        ctx.synchronize()
        out[direction] = (L.cpu().numpy(), Cv.cpu().numpy())
    return cl.cpu().numpy().view(np.uint32), cr.cpu().numpy().view(np.uint32), out


@pytest.mark.parametrize("W,H,D,F,seed", [(320, 240, 128, 1, 1), (131, 77, 64, 2, 2), (70, 9, 128, 1, 3), (257, 33, 100, 3, 4), (9, 7, 8, 1, 5)])
def test_census_and_all_eight_paths_match_the_oracle(W, H, D, F, seed):
    from oracle import pysgm
    from oracle import sgm_numpy as sn
    pairs = [sn.make_stereo(W, H, seed * 10 + f, D, n_boxes=3) for f in range(F)]
    left, right = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    ctx = _ctx(W, H, F)
    cl, cr, out = _stages(ctx, left, right, D)
    for f in range(F):
        assert np.array_equal(cl[f], pysgm.census(left[f])) and np.array_equal(cr[f], pysgm.census(right[f]))
        Cref = pysgm.cost(cl[f], cr[f], D)
        for direction in range(8):
            L, Cv = out[direction]
            assert np.array_equal(Cv[f], Cref), (f, direction)
            assert np.array_equal(L[f], pysgm.aggregate(Cref, 6, 96, direction)), (f, direction)
    ctx.close()


def test_fixture_and_argument_checks():
    from moving_object_detector_amd import capi
    g = np.load(GOLD)
    D = int(g["D"])
    ctx = _ctx(320, 240, 1)
    cl, cr, out = _stages(ctx, g["left"][None], g["right"][None], D, int(g["P1"]), int(g["P2"]))
    assert np.array_equal(cl[0], g["census_left"]) and np.array_equal(cr[0], g["census_right"])
    for direction, k in ((i, f"path{i}") for i in range(8)):        # all eight aggregation paths are pinned
        L, Cv = out[direction]
        assert np.array_equal(Cv[0][g["rows"]], g["cost_rows"])
        assert np.array_equal(L[0][g["rows"]], g[k + "_rows"]) and np.array_equal(L[0].astype(np.uint32).sum(axis=2), g[k + "_sum"])
    # other penalties
    from oracle import pysgm
    prm = capi.ModSgmParams(64, 3, 40, 8, 1, 1)
    dev = ctx.device
    tl, tr = torch.from_numpy(cl.view(np.int32)).to(dev), torch.from_numpy(cr.view(np.int32)).to(dev)
    L = torch.empty((1, 240, 320, 64), dtype=torch.uint8, device=dev)
    assert ctx.lib.mod_sgm_path_dev(ctx.h, 1, tl.data_ptr(), tr.data_ptr(), C.byref(prm), 0, L.data_ptr(), None) == 0
    ctx.synchronize()
    assert np.array_equal(L.cpu().numpy()[0], pysgm.aggregate(pysgm.cost(cl[0], cr[0], 64), 3, 40, 0))
    # what cannot work is an error — never a silent no-op
    for bad_dir in (8, -1):
        assert ctx.lib.mod_sgm_path_dev(ctx.h, 1, tl.data_ptr(), tr.data_ptr(), C.byref(prm), bad_dir, L.data_ptr(), None) == capi.MOD_ERR_INVALID_ARGUMENT
    assert b"direction" in ctx.lib.mod_last_error(ctx.h)
    for bad in (capi.ModSgmParams(0, 6, 96, 8, 1, 1), capi.ModSgmParams(129, 6, 96, 8, 1, 1), capi.ModSgmParams(64, 6, 230, 8, 1, 1), capi.ModSgmParams(64, 50, 40, 8, 1, 1),
                capi.ModSgmParams(64, 6, 96, 5, 1, 1)):
        assert ctx.lib.mod_sgm_path_dev(ctx.h, 1, tl.data_ptr(), tr.data_ptr(), C.byref(bad), 0, L.data_ptr(), None) == capi.MOD_ERR_INVALID_ARGUMENT
    assert ctx.lib.mod_sgm_census_dev(ctx.h, 1, None, tl.data_ptr()) == capi.MOD_ERR_INVALID_ARGUMENT
    ctx.close()


def _compute(ctx, left, right, **kw):
    from moving_object_detector_amd import capi
    F, H, W = left.shape
    dev = ctx.device
    prm = capi.ModSgmParams(kw.get("D", 128), kw.get("P1", 6), kw.get("P2", 96), kw.get("paths", 8), int(kw.get("lr_check", True)),
                            int(kw.get("median", True)))
    out = torch.full((F, H, W), -7.0, dtype=torch.float32, device=dev)
    tl, tr = torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev)       # kept alive until the kernels have run
    rc = ctx.lib.mod_sgm_compute_dev(ctx.h, F, tl.data_ptr(), tr.data_ptr(), C.byref(prm), out.data_ptr())
    assert rc == 0, ctx.lib.mod_last_error(ctx.h)
    ctx.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("W,H,D,F,seed", [(320, 240, 128, 2, 1), (131, 77, 64, 1, 2), (70, 9, 128, 1, 3), (9, 7, 8, 1, 5), (96, 40, 33, 11, 6),
                                          (64, 24, 16, 20, 8)])     # 20 frames: three groups, so a volume set is used twice
def test_complete_estimator_matches_the_oracle(W, H, D, F, seed):
    from oracle import pysgm
    from oracle import sgm_numpy as sn
    pairs = [sn.make_stereo(W, H, seed * 10 + f, D, n_boxes=3) for f in range(F)]
    left, right = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    ctx = _ctx(W, H, F)
    for kw in (dict(), dict(paths=4), dict(lr_check=False, median=False), dict(P1=3, P2=40, median=False)):
        got = _compute(ctx, left, right, D=D, **kw)
        for f in range(F):
            want = pysgm.compute(left[f], right[f], D, kw.get("P1", 6), kw.get("P2", 96), kw.get("paths", 8), kw.get("lr_check", True),
                                 kw.get("median", True))
            assert np.array_equal(got[f], want), (kw, f, int((got[f] != want).sum()))
    g = np.load(GOLD)
    if (W, H, D) == (320, 240, 128):
        assert np.array_equal(_compute(ctx, g["left"][None], g["right"][None])[0], g["disparity"])
        # the host-pointer form
        from moving_object_detector_amd import capi
        gl, gr = np.ascontiguousarray(g["left"]), np.ascontiguousarray(g["right"])
        host = np.full((H, W), -7.0, np.float32)
        prm = capi.ModSgmParams(128, 6, 96, 8, 1, 1)
        assert ctx.lib.mod_sgm_compute_host(ctx.h, gl.ctypes.data, gr.ctypes.data, C.byref(prm), host.ctypes.data) == 0
        assert np.array_equal(host, g["disparity"])
        assert ctx.lib.mod_sgm_compute_host(ctx.h, None, gr.ctypes.data, C.byref(prm), host.ctypes.data) == capi.MOD_SKIP_NO_DISPARITY_NOW
    ctx.close()


def test_scratch_regrows_between_calls():
    """One context, calls with growing then shrinking disparity ranges and frame counts: the estimator's scratch (two sets of census
    planes and cost volumes, sized on first use) is re-allocated when a call needs more and reused otherwise."""
    from oracle import pysgm
    from oracle import sgm_numpy as sn
    W, H = 80, 36
    ctx = _ctx(W, H, 20)
    # the tail alternates (many disparities, few frames) with (few, many): the scratch is grow-only in both dimensions
    for D, F, seed in ((16, 2, 1), (64, 9, 2), (128, 3, 3), (32, 20, 4), (128, 17, 5), (128, 2, 6), (16, 20, 7), (128, 2, 8), (16, 20, 9)):
        pairs = [sn.make_stereo(W, H, seed * 100 + f, D, n_boxes=2) for f in range(F)]
        left, right = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
        got = _compute(ctx, left, right, D=D)
        for f in (0, F // 2, F - 1):
            assert np.array_equal(got[f], pysgm.compute(left[f], right[f], D, 6, 96, 8, True, True)), (D, F, f)
    ctx.close()


def test_config5_images_to_moving_objects(oracle):
    """BASELINE config 5's data path at its stated size: 1920 x 1080 stereo images -> on-GPU SGM disparity (now and previous) ->
    scene flow + clustering, every stage against its CPU restatement.  The disparity planes never leave the GPU."""
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.pipeline import PLANES, Context
    from oracle import pysgm
    from oracle import sgm_numpy as sn
    from util import bits_equal
    W, H, D = 1920, 1080, 128
    l0, r0, truth0 = sn.make_stereo(W, H, 41, D, n_boxes=5)
    l1, r1, _ = sn.make_stereo(W, H, 42, D, n_boxes=5)
    cam = synth.make_camera(W, H)
    cam.min_disparity, cam.max_disparity = np.float32(0.0), np.float32(D - 1)      # the DisparityImage fields of this estimator
    prm = synth.Params()
    ctx = Context(W, H, max_frames=2)
    ctx.set_camera(cam)
    ctx.set_params(prm)
    dev = ctx.device
    sp = capi.ModSgmParams(D, 6, 96, 8, 1, 1)
    imgs_l, imgs_r = torch.from_numpy(np.stack([l0, l1])).to(dev), torch.from_numpy(np.stack([r0, r1])).to(dev)
    disp = torch.empty((2, H, W), dtype=torch.float32, device=dev)                 # [0] = previous, [1] = now
    assert ctx.lib.mod_sgm_compute_dev(ctx.h, 2, imgs_l.data_ptr(), imgs_r.data_ptr(), C.byref(sp), disp.data_ptr()) == 0
    ctx.synchronize()
    want = [pysgm.compute(l0, r0, D), pysgm.compute(l1, r1, D)]
    got = disp.cpu().numpy()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert (got >= 0).mean() > 0.6
    # scene flow + clusters from the device-resident disparities
    rng = np.random.default_rng(5)
    flow = rng.uniform(-3, 3, size=(1, H, W, 2)).astype(np.float32)
    t, q = np.array([[0.01, 0.0, 0.08]]), np.array([[0.0, 0.003, 0.0, 1.0]])
    ws = ctx.workspace(1)
    b = ctx.make_batch(disp[1:2], disp[0:1], torch.from_numpy(flow).to(dev), t, q, [0.1])
    assert ctx.process(b, ws) == 0
    ctx.synchronize()
    ref = oracle.construct(cam, prm, want[1], want[0], flow[0], t[0], q[0], 0.1, "tidy")
    for i, k in enumerate(PLANES):
        assert bits_equal(ws["planes"][i, 0].cpu().numpy(), ref[k]), k
    labels, objs, K = oracle.cluster(ref, prm, "tidy")
    assert np.array_equal(ws["labels"][0].cpu().numpy(), labels) and int(ws["n_objects"][0]) == len(objs)
    # a stream with something to find: the camera stands still on image pair 0 (previous = now, identity ego-motion) and the flow
    # moves the boxes by 24 px, so the clusterer has moving objects at 1080p and the default parameters
    from util import compare_objects
    flow2 = synth.make_box_flow(truth0, shift=24.0)[None]
    t2, q2 = np.zeros((1, 3)), np.array([[0.0, 0.0, 0.0, 1.0]])
    b2 = ctx.make_batch(disp[0:1], disp[0:1], torch.from_numpy(flow2).to(dev), t2, q2, [1.0 / 15.0])
    assert ctx.process(b2, ws) == 0
    ctx.synchronize()
    ref2 = oracle.construct(cam, prm, want[0], want[0], flow2[0], t2[0], q2[0], 1.0 / 15.0, "tidy")
    for i, k in enumerate(PLANES):
        assert bits_equal(ws["planes"][i, 0].cpu().numpy(), ref2[k]), k
    labels2, objs2, K2 = oracle.cluster(ref2, prm, "tidy")
    n_objects = int(ws["n_objects"][0])
    assert n_objects > 0 and n_objects == len(objs2), (n_objects, len(objs2))
    assert np.array_equal(ws["labels"][0].cpu().numpy(), labels2)
    compare_objects(ctx.objects_to_host(ws)[0], objs2, strict_velocity=True)
    ctx.close()


def test_sgm_soak_short():
    """Random sizes (ragged and multiple-of-4: both variants of the four-line path kernels), disparity counts, penalties, path counts and
    frame counts for 15 s against the CPU restatement (tools/soak_sgm.py; 2 878 configurations passed in the round's 120 s run)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_sgm.py"), "15", "11"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sgm soak passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert sum(1 for l in r.stdout.splitlines() if " D=128 " in l) >= 20
