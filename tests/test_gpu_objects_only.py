"""GPU: output gating of the fused call (SURVEY.md 8(b) "optional outputs requested by flags").  The reference builds and ships the
cloud only while ~scene_flow has subscribers (scene_flow_constructor.cpp:141-142) and renders its cluster image only for subscribers
(clusterer_nodelet.cpp:233-238); mod_process_dev with ModSceneFlowPlanes.x = .y = NULL (and no labels plane) is the call of a node
that serves ~moving_objects alone.  Its objects must be the six-plane call's objects byte for byte."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _run(ctx, batch, frames, **kw):
    ws = ctx.workspace(frames, **kw)
    ws["planes"].fill_(-12345.0)
    ws["objects"].zero_()
    assert ctx.process(batch, ws) == 0
    ctx.synchronize()
    return ws


@pytest.mark.parametrize("W,H,F", [(1280, 720, 4), (1242, 376, 3), (321, 200, 2)])
def test_objects_only_equals_the_six_plane_call(oracle, W, H, F):
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.pipeline import Context
    from util import compare_objects
    cam, sq = synth.make_sequence(W, H, F, seed=4)
    prm = synth.Params() if W >= 1242 else synth.Params(dynamic_flow_diff=1, cluster_size=60)
    ctx = Context(W, H, max_frames=F, max_objects=W * H // prm.cluster_size + 1)
    ctx.set_camera(cam); ctx.set_params(prm)
    dev = ctx.device
    d = torch.from_numpy(sq["disparity"]).to(dev)
    batch = ctx.make_batch(d[1:].contiguous(), d[:-1].contiguous(), torch.from_numpy(sq["flow"]).to(dev), sq["t"], sq["q"], sq["dt"])
    full = _run(ctx, batch, F)
    planes6 = full["planes"].cpu().numpy().copy()
    labels6 = full["labels"].cpu().numpy().copy()
    n6 = full["n_objects"].cpu().numpy().copy()
    raw6 = full["objects"].cpu().numpy().copy()
    objs6 = ctx.objects_to_host(full)
    assert int(n6.sum()) > 0
    only = _run(ctx, batch, F, labels=False, xy=False)
    planes4 = only["planes"].cpu().numpy()
    assert (planes4[0] == -12345.0).all() and (planes4[1] == -12345.0).all(), "x / y planes were written although none was passed"
    for i in (2, 3, 4, 5):                               # z, vx, vy, vz: the same bits
        assert np.array_equal(planes4[i].view(np.uint32), planes6[i].view(np.uint32)), i
    n4 = only["n_objects"].cpu().numpy()
    assert np.array_equal(n4, n6)
    raw4 = only["objects"].cpu().numpy()
    for f in range(F):
        assert raw4[f, : n4[f]].tobytes() == raw6[f, : n6[f]].tobytes(), f
    # x, y only: labels still come out when asked for
    lab = _run(ctx, batch, F, labels=True, xy=False)
    assert np.array_equal(lab["labels"].cpu().numpy(), labels6)
    # and the six-plane call equals the oracle (so does, by the above, the objects-only call)
    host = {"disparity_now": sq["disparity"][1:], "disparity_prev": sq["disparity"][:-1]}
    for f in range(min(F, 2)):
        ref = oracle.construct(cam, prm, host["disparity_now"][f], host["disparity_prev"][f], sq["flow"][f], sq["t"][f], sq["q"][f], float(sq["dt"][f]), "tidy")
        rl, ro, _ = oracle.cluster(ref, prm, "tidy", max_objects=W * H)
        assert np.array_equal(labels6[f], rl)
        compare_objects(objs6[f], ro, strict_velocity=True)
    # one of x, y alone is an argument error; so is a missing x / y where the planes ARE the product (mod_scene_flow_dev, mod_cluster_dev)
    s = ctx._planes_struct(full)
    s.x = None
    out = ctx._cluster_struct(full)
    assert ctx.lib.mod_process_dev(ctx.h, C.byref(batch), C.byref(s), C.byref(out)) == capi.MOD_ERR_INVALID_ARGUMENT
    s.y = None
    assert ctx.lib.mod_process_dev(ctx.h, C.byref(batch), C.byref(s), C.byref(out)) == 0
    assert ctx.lib.mod_scene_flow_dev(ctx.h, C.byref(batch), C.byref(s)) == capi.MOD_ERR_INVALID_ARGUMENT
    assert ctx.lib.mod_cluster_dev(ctx.h, F, C.byref(s), C.byref(out)) == capi.MOD_ERR_INVALID_ARGUMENT
    ctx.synchronize()
    ctx.close()
