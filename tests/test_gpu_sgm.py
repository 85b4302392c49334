"""GPU: the first on-GPU disparity stages (census transform, the two horizontal aggregation paths with the matching cost computed
on the fly) through the C ABI, bit for bit against the CPU restatement and the committed fixture."""
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
    for direction in (0, 1):
        L = torch.full((F, H, W, D), 255, dtype=torch.uint8, device=dev)
        Cv = torch.full((F, H, W, D), 255, dtype=torch.uint8, device=dev)
        assert ctx.lib.mod_sgm_path_dev(ctx.h, F, cl.data_ptr(), cr.data_ptr(), C.byref(prm), direction, L.data_ptr(), Cv.data_ptr()) == 0
        ctx.synchronize()
        out[direction] = (L.cpu().numpy(), Cv.cpu().numpy())
    return cl.cpu().numpy().view(np.uint32), cr.cpu().numpy().view(np.uint32), out


@pytest.mark.parametrize("W,H,D,F,seed", [(320, 240, 128, 1, 1), (131, 77, 64, 2, 2), (70, 9, 128, 1, 3), (257, 33, 100, 3, 4), (9, 7, 8, 1, 5)])
def test_census_and_horizontal_paths_match_the_oracle(W, H, D, F, seed):
    from oracle import pysgm
    from oracle import sgm_numpy as sn
    pairs = [sn.make_stereo(W, H, seed * 10 + f, D, n_boxes=3) for f in range(F)]
    left, right = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    ctx = _ctx(W, H, F)
    cl, cr, out = _stages(ctx, left, right, D)
    for f in range(F):
        assert np.array_equal(cl[f], pysgm.census(left[f])) and np.array_equal(cr[f], pysgm.census(right[f]))
        Cref = pysgm.cost(cl[f], cr[f], D)
        for direction in (0, 1):
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
    for direction, k in ((0, "path0"), (1, "path1")):
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
    # what does not exist yet, and what cannot work, is an error — never a silent no-op
    for bad_dir in (2, 7, -1):
        assert ctx.lib.mod_sgm_path_dev(ctx.h, 1, tl.data_ptr(), tr.data_ptr(), C.byref(prm), bad_dir, L.data_ptr(), None) == capi.MOD_ERR_INVALID_ARGUMENT
    assert b"horizontal" in ctx.lib.mod_last_error(ctx.h)
    for bad in (capi.ModSgmParams(0, 6, 96, 8, 1, 1), capi.ModSgmParams(129, 6, 96, 8, 1, 1), capi.ModSgmParams(64, 6, 230, 8, 1, 1), capi.ModSgmParams(64, 50, 40, 8, 1, 1)):
        assert ctx.lib.mod_sgm_path_dev(ctx.h, 1, tl.data_ptr(), tr.data_ptr(), C.byref(bad), 0, L.data_ptr(), None) == capi.MOD_ERR_INVALID_ARGUMENT
    assert ctx.lib.mod_sgm_census_dev(ctx.h, 1, None, tl.data_ptr()) == capi.MOD_ERR_INVALID_ARGUMENT
    ctx.close()
