"""GPU, BASELINE.json's full sizes: size-independent properties of the hot path (the oracle is too slow to run on
every full-size frame, so these complement the small-size parity tests)."""
import numpy as np
import pytest

from util import PLANES, bits_equal

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _run(W, H, batch, prm, order=None):
    from moving_object_detector_amd.pipeline import Context
    F = batch["disparity_now"].shape[0]
    idx = list(range(F)) if order is None else order
    ctx = Context(W, H, max_frames=len(idx))
    ctx.set_camera(__import__("moving_object_detector_amd.synth", fromlist=["x"]).make_camera(W, H))
    ctx.set_params(prm)
    ws = ctx.workspace(len(idx), aos=True)
    dev = ctx.device
    b = ctx.make_batch(torch.from_numpy(batch["disparity_now"][idx]).to(dev), torch.from_numpy(batch["disparity_prev"][idx]).to(dev),
                       torch.from_numpy(batch["flow"][idx]).to(dev), batch["t"][idx], batch["q"][idx], batch["dt"][idx])
    assert ctx.process(b, ws) == 0
    ctx.synchronize()
    out = {k: ws["planes"][i].cpu().numpy() for i, k in enumerate(PLANES)}
    out["labels"] = ws["labels"].cpu().numpy()
    out["mask"] = ws["mask"].cpu().numpy()
    out["aos"] = ws["aos"].cpu().numpy()
    out["objects"] = ctx.objects_to_host(ws)
    out["n_clusters"] = ws["n_clusters"].cpu().numpy()
    # pack / unpack round trip on the device (pcl::toROSMsg / fromROSMsg payloads)
    import ctypes as C
    planes2 = torch.zeros_like(ws["planes"])
    from moving_object_detector_amd import capi
    s = capi.ModSceneFlowPlanes()
    for i, k in enumerate(PLANES):
        setattr(s, k, planes2[i].data_ptr())
    assert ctx.lib.mod_unpack_cloud_dev(ctx.h, len(idx), ws["aos"].data_ptr(), C.byref(s)) == 0
    ctx.synchronize()
    out["roundtrip"] = planes2.cpu().numpy()
    ctx.close()
    return out


@pytest.mark.parametrize("W,H", [(1280, 720), (1920, 1080)])
def test_full_size_properties(W, H):
    from moving_object_detector_amd import synth
    from oracle import numpy_ref
    cam, batch = synth.make_batch(W, H, 2, seed=31)
    prm = synth.Params()
    # the same pair twice in one batch, in swapped order in a second run
    dup = {k: np.concatenate([v, v]) for k, v in batch.items()}
    a = _run(W, H, dup, prm)
    b = _run(W, H, dup, prm, order=[1, 0, 3, 2])
    for k in PLANES + ("labels", "mask"):
        # determinism: identical frames give identical bytes wherever they sit in the batch / whichever blocks ran them
        assert np.array_equal(a[k][0].view(np.uint32), a[k][2].view(np.uint32)), k
        assert np.array_equal(a[k][1].view(np.uint32), b[k][0].view(np.uint32)), k
    for f in range(2):
        lab = a["labels"][f]
        K = int(a["n_clusters"][f])
        # labels are -1 or 0..K-1, every label present, every cluster at least cluster_size pixels
        assert lab.min() >= -1 and lab.max() == K - 1
        sizes = np.bincount(lab[lab >= 0], minlength=K)
        assert (sizes >= prm.cluster_size).all()
        assert [int(o["n_points"]) for o in a["objects"][f]] == sizes.tolist()
        # only dynamic pixels carry labels, and the mask equals calculateDynamicMap of the velocity planes
        dyn = numpy_ref.dynamic_mask(prm, a["vx"][f], a["vy"][f], a["vz"][f])
        bits = np.unpackbits(a["mask"][f].view(np.uint8), axis=-1, bitorder="little")[:, :W].astype(bool)
        assert np.array_equal(bits, dyn)
        assert not (lab[~dyn] >= 0).any()
        # first_edge_key order: the first pixel (raster order) of cluster k precedes... at least labels appear sorted by
        # the raster position of their first up-left-edge pixel; a cheap necessary condition is that label 0's first pixel
        # comes no later than any other cluster's second row
        # bbox / centre follow from the member coordinates exactly (float32 min/max, (min+max)/2)
        for o in a["objects"][f]:
            m = lab == int(o["id"])
            mn = np.array([a[k][f][m].min() for k in ("x", "y", "z")], np.float32)
            mx = np.array([a[k][f][m].max() for k in ("x", "y", "z")], np.float32)
            assert np.array_equal(o["bounding_box"], (mx - mn).astype(np.float64))
            assert np.array_equal(o["center"], ((mn + mx) / np.float32(2)).astype(np.float64))
            # the reported velocity belongs to a member whose ||v|| is the element at size/2 of the descending order
            nr = numpy_ref.norm3(a["vx"][f][m], a["vy"][f][m], a["vz"][f][m])
            med = np.sort(nr)[::-1][nr.size // 2]
            v = o["velocity"].astype(np.float32)
            assert numpy_ref.norm3(v[0:1], v[1:2], v[2:3])[0] == med
        # AoS output == SoA planes, and unpack(pack) is the identity
        for j, k in zip((0, 1, 2, 4, 5, 6), PLANES):
            assert bits_equal(a["aos"][f][..., j], a[k][f])
            assert bits_equal(a["roundtrip"][["x", "y", "z", "vx", "vy", "vz"].index(k)][f], a[k][f])


def test_full_size_frame_matches_oracle_1920x1080(oracle):
    from moving_object_detector_amd import synth
    from test_gpu_parity import _check_against_oracle, _run_gpu
    cam, batch = synth.make_batch(1920, 1080, 1, seed=41)
    prm = synth.Params()
    out = _run_gpu(cam, prm, batch)
    _check_against_oracle(oracle, cam, prm, batch, out)


def test_stage_timers_follow_the_stage_mask():
    """mod_set_profiling(stage_mask): only the selected kernel groups are bracketed; results do not depend on it."""
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.pipeline import Context
    W, H, F = 320, 240, 2
    cam, batch = synth.make_batch(W, H, F, seed=2)
    ctx = Context(W, H, max_frames=F)
    ctx.set_camera(cam)
    ctx.set_params(synth.Params(cluster_size=200))
    ws = ctx.workspace(F)
    dev = ctx.device
    b = ctx.make_batch(torch.from_numpy(batch["disparity_now"]).to(dev), torch.from_numpy(batch["disparity_prev"]).to(dev),
                       torch.from_numpy(batch["flow"]).to(dev), batch["t"], batch["q"], batch["dt"])
    assert ctx.process(b, ws) == 0
    ctx.synchronize()
    ref_labels = ws["labels"].cpu().numpy().copy()
    ctx.set_profiling(True, stages=[capi.MOD_STAGE_SCENE_FLOW, capi.MOD_STAGE_FINAL])
    ctx.reset_stage_times()
    for _ in range(3):
        assert ctx.process(b, ws) == 0
    times = [ctx.stage_time(i) for i in range(capi.MOD_STAGE_COUNT)]
    for i, (ms, calls) in enumerate(times):
        if i in (capi.MOD_STAGE_SCENE_FLOW, capi.MOD_STAGE_FINAL):
            assert calls == 3 and ms > 0.0
        else:
            assert calls == 0 and ms == 0.0
    ctx.set_profiling(True)
    ctx.reset_stage_times()
    assert ctx.process(b, ws) == 0
    assert all(ctx.stage_time(i)[1] == 1 for i in range(capi.MOD_STAGE_COUNT))
    ctx.set_profiling(False)
    assert ctx.lib.mod_set_profiling(ctx.h, 0x80) == capi.MOD_ERR_INVALID_ARGUMENT      # not a stage bit
    assert np.array_equal(ws["labels"].cpu().numpy(), ref_labels)
    ctx.close()
