"""GPU: the HIP path through the C ABI against the committed golden fixtures — device entry points, the host-pointer
entry points a ROS node would call, skip / error codes."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from test_oracle_golden import GOLD, golden_objects, load_case
from util import PLANES, bits_equal, compare_objects, first_mismatch

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _ctx(cam, prm, frames=1):
    from moving_object_detector_amd.pipeline import Context
    ctx = Context(cam.width, cam.height, max_frames=frames, max_objects=cam.width * cam.height // prm.cluster_size + 1)
    ctx.set_camera(cam)
    ctx.set_params(prm)
    return ctx


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_device_path_matches_golden(path):
    g, cam, prm = load_case(path)
    ctx = _ctx(cam, prm)
    ws = ctx.workspace(1, aos=True, extras=True)
    dev = ctx.device
    b = ctx.make_batch(torch.from_numpy(g["d_now"][None]).to(dev), torch.from_numpy(g["d_prev"][None]).to(dev),
                       torch.from_numpy(g["flow"][None]).to(dev), g["t"][None], g["q"][None], [float(np.asarray(g["dt"]).item())])
    assert ctx.process(b, ws) == 0
    ctx.synchronize()
    for i, k in enumerate(PLANES):
        got = ws["planes"][i, 0].cpu().numpy()
        assert bits_equal(got, g[k]), (k, first_mismatch(got, g[k]))
    assert bits_equal(ws["static_flow"][0].cpu().numpy(), g["static_flow"])
    assert bits_equal(ws["depth"][0].cpu().numpy(), g["depth"])
    assert np.array_equal(ws["labels"][0].cpu().numpy(), g["labels"])
    assert int(ws["n_clusters"][0]) == int(np.asarray(g["K"]).item())
    compare_objects(ctx.objects_to_host(ws)[0], golden_objects(g), strict_velocity=False)
    ctx.close()


@pytest.mark.parametrize("path", GOLD[:3], ids=[os.path.basename(p) for p in GOLD[:3]])
def test_host_entry_points_match_golden(path):
    """mod_process_frame_host + mod_cluster_cloud_host: what a ROS node with host-side messages calls."""
    from moving_object_detector_amd import capi
    from moving_object_detector_amd.pipeline import OBJECT_DTYPE
    g, cam, prm = load_case(path)
    g = {k: np.ascontiguousarray(g[k]) for k in g.files}     # NpzFile hands out temporaries: keep the arrays alive
    ctx = _ctx(cam, prm)
    H, W = g["d_now"].shape
    cloud = np.zeros((H, W, 8), np.float32)
    labels = np.zeros((H, W), np.int32)
    objs = np.zeros(64, OBJECT_DTYPE)
    n = C.c_int32(-1)
    tf = capi.transforms_array([g["t"]], [g["q"]])
    rc = ctx.lib.mod_process_frame_host(ctx.h, g["d_now"].ctypes.data, g["d_prev"].ctypes.data, g["flow"].ctypes.data, tf,
                                        float(np.asarray(g["dt"]).item()), cloud.ctypes.data, labels.ctypes.data, objs.ctypes.data, 64, C.byref(n))
    assert rc == 0
    for j, k in zip((0, 1, 2, 4, 5, 6), PLANES):
        assert bits_equal(cloud[..., j], g[k]), k
    assert np.array_equal(labels, g["labels"]) and n.value == int(np.asarray(g["n_objects"]).item())
    compare_objects(objs[: n.value], golden_objects(g), strict_velocity=False)
    # ~synthetic_optical_flow for a node with host buffers (calculateStaticOpticalFlow, scene_flow_constructor.cpp:65-89,141-145)
    sflow = np.full((H, W, 2), 7.0, np.float32)
    assert ctx.lib.mod_static_flow_host(ctx.h, g["d_prev"].ctypes.data, tf, sflow.ctypes.data) == 0
    assert bits_equal(sflow, g["static_flow"])
    assert ctx.lib.mod_static_flow_host(ctx.h, None, tf, sflow.ctypes.data) == capi.MOD_SKIP_NO_DISPARITY_PREV
    assert ctx.lib.mod_static_flow_host(ctx.h, g["d_prev"].ctypes.data, None, sflow.ctypes.data) == capi.MOD_SKIP_NO_TRANSFORM
    # clusterer alone on the PointCloud2 payload (ClustererNodelet::dataCB)
    labels2 = np.zeros((H, W), np.int32)
    n2 = C.c_int32(-1)
    rc = ctx.lib.mod_cluster_cloud_host(ctx.h, cloud.ctypes.data, W, H, 32, 32 * W, labels2.ctypes.data, objs.ctypes.data, 64, C.byref(n2))
    assert rc == 0 and np.array_equal(labels2, g["labels"]) and n2.value == n.value
    # without a labels plane (no subscriber of the cluster image, clusterer_nodelet.cpp:235-236): the same objects
    objs3, n3 = np.zeros(64, OBJECT_DTYPE), C.c_int32(-1)
    rc = ctx.lib.mod_process_frame_host(ctx.h, g["d_now"].ctypes.data, g["d_prev"].ctypes.data, g["flow"].ctypes.data, tf,
                                        float(np.asarray(g["dt"]).item()), None, None, objs3.ctypes.data, 64, C.byref(n3))
    assert rc == 0 and n3.value == n.value and objs3[: n3.value].tobytes() == objs[: n.value].tobytes()
    rc = ctx.lib.mod_cluster_cloud_host(ctx.h, cloud.ctypes.data, W, H, 32, 32 * W, None, objs3.ctypes.data, 64, C.byref(n3))
    assert rc == 0 and n3.value == n.value and objs3[: n3.value].tobytes() == objs[: n.value].tobytes()
    # a caller that only counts the objects (n_objects alone) still gets the real count: the cluster stage runs
    n5 = C.c_int32(-1)
    rc = ctx.lib.mod_process_frame_host(ctx.h, g["d_now"].ctypes.data, g["d_prev"].ctypes.data, g["flow"].ctypes.data, tf,
                                        float(np.asarray(g["dt"]).item()), None, None, None, 0, C.byref(n5))
    assert rc == 0 and n5.value == n.value
    # ... and with no cluster output at all (not even the count) only the scene-flow stage runs: the cloud is the same
    cloud5 = np.zeros((H, W, 8), np.float32)
    rc = ctx.lib.mod_process_frame_host(ctx.h, g["d_now"].ctypes.data, g["d_prev"].ctypes.data, g["flow"].ctypes.data, tf,
                                        float(np.asarray(g["dt"]).item()), cloud5.ctypes.data, None, None, 0, None)
    assert rc == 0 and cloud5.tobytes() == cloud.tobytes()
    # a clusterer-only context (the nodelet in its own process): no camera is ever set, the size comes with the cloud
    from moving_object_detector_amd.pipeline import Context
    solo = Context(W + 7, H + 3, max_frames=1, max_objects=(W + 7) * (H + 3) // prm.cluster_size + 1)
    assert solo.lib.mod_cluster_cloud_host(solo.h, cloud.ctypes.data, W, H, 32, 32 * W, labels2.ctypes.data, objs3.ctypes.data, 64, C.byref(n3)) == capi.MOD_ERR_NOT_CONFIGURED
    solo.set_params(prm)
    labels4 = np.full((H, W), -5, np.int32)
    rc = solo.lib.mod_cluster_cloud_host(solo.h, cloud.ctypes.data, W, H, 32, 32 * W, labels4.ctypes.data, objs3.ctypes.data, 64, C.byref(n3))
    assert rc == 0 and np.array_equal(labels4, g["labels"]) and n3.value == n.value and objs3[: n3.value].tobytes() == objs[: n.value].tobytes()
    assert solo.lib.mod_cluster_cloud_host(solo.h, cloud.ctypes.data, W + 8, H, 32, 32 * W, None, objs3.ctypes.data, 64, C.byref(n3)) == capi.MOD_ERR_CAPACITY
    solo.close()
    # a cloud of the wrong size is an error, not a silent skip
    rc = ctx.lib.mod_cluster_cloud_host(ctx.h, cloud.ctypes.data, W - 1, H, 32, 32 * W, labels2.ctypes.data, objs.ctypes.data, 64, C.byref(n2))
    assert rc == capi.MOD_ERR_INVALID_ARGUMENT and b"cloud size" in ctx.lib.mod_last_error(ctx.h)
    ctx.close()


def test_missing_inputs_return_skip_codes():
    """construct() guards (scene_flow_constructor.cpp:104,110,122,127,133): a missing input is a skip, not an error."""
    from moving_object_detector_amd import capi
    g, cam, prm = load_case(GOLD[0])
    g = {k: np.ascontiguousarray(g[k]) for k in g.files}
    ctx = _ctx(cam, prm)
    n = C.c_int32(-1)
    tf = capi.transforms_array([g["t"]], [g["q"]])
    a = (g["d_now"].ctypes.data, g["d_prev"].ctypes.data, g["flow"].ctypes.data)
    call = lambda dn, dp, fl, t: ctx.lib.mod_process_frame_host(ctx.h, dn, dp, fl, t, 0.1, None, None, None, 0, C.byref(n))
    assert call(a[0], a[1], None, tf) == capi.MOD_SKIP_NO_FLOW and n.value == 0
    assert call(a[0], None, a[2], tf) == capi.MOD_SKIP_NO_DISPARITY_PREV
    assert call(a[0], a[1], a[2], None) == capi.MOD_SKIP_NO_TRANSFORM
    assert call(None, a[1], a[2], tf) == capi.MOD_SKIP_NO_DISPARITY_NOW
    assert call(a[0], a[1], a[2], tf) == 0
    ctx.close()


@pytest.mark.parametrize("path", GOLD[:2], ids=[os.path.basename(p) for p in GOLD[:2]])
def test_depth_is_published_on_skipped_frames(path):
    """construct() publishes ~depth whenever disparity_now exists, before the guards that end a frame without scene flow
    (scene_flow_constructor.cpp:110-123): frame 0 of every stream (no flow, no previous disparity) and every frame after an
    estimator failure still get their depth image; nothing else is written."""
    from moving_object_detector_amd import capi
    g, cam, prm = load_case(path)
    g = {k: np.ascontiguousarray(g[k]) for k in g.files}
    ctx = _ctx(cam, prm)
    dev = ctx.device
    dn, dp, fl = (torch.from_numpy(g[k][None]).to(dev) for k in ("d_now", "d_prev", "flow"))
    tq = (g["t"][None], g["q"][None], [0.1])
    cases = [(dn, dp, None, tq, capi.MOD_SKIP_NO_FLOW), (dn, None, fl, tq, capi.MOD_SKIP_NO_DISPARITY_PREV),
             (dn, dp, fl, (None, None, None), capi.MOD_SKIP_NO_TRANSFORM), (dn, None, None, (None, None, None), capi.MOD_SKIP_NO_FLOW)]
    for fused in (True, False):
        for a, b_, c_, (t, q, dt), code in cases:
            ws = ctx.workspace(1, aos=True, extras=True)
            ws["depth"].fill_(-7.0)
            ws["planes"].fill_(-7.0)
            ws["labels"].fill_(-7)
            b = ctx.make_batch(a, b_, c_, t, q, dt)
            rc = ctx.process(b, ws) if fused else ctx.scene_flow(b, ws)
            assert rc == code
            ctx.synchronize()
            assert bits_equal(ws["depth"][0].cpu().numpy(), g["depth"])
            assert bool((ws["planes"] == -7.0).all()) and bool((ws["labels"] == -7).all())      # nothing else is published
    # no disparity_now: nothing at all
    ws["depth"].fill_(-7.0)
    assert ctx.process(ctx.make_batch(dn, dp, fl, *tq), ws) == 0          # (sanity: the complete frame runs)
    b = ctx.make_batch(dn, dp, fl, *tq)
    b.disparity_now = None
    ws["depth"].fill_(-7.0)
    assert ctx.process(b, ws) == capi.MOD_SKIP_NO_DISPARITY_NOW
    ctx.synchronize()
    assert bool((ws["depth"] == -7.0).all())
    # the stand-alone forms: device planes and one host frame
    out = torch.empty_like(dn)
    assert ctx.lib.mod_depth_image_dev(ctx.h, 1, dn.data_ptr(), out.data_ptr()) == 0
    ctx.synchronize()
    assert bits_equal(out[0].cpu().numpy(), g["depth"])
    host = np.full(g["d_now"].shape, -7.0, np.float32)
    assert ctx.lib.mod_depth_image_host(ctx.h, g["d_now"].ctypes.data, host.ctypes.data) == 0
    assert bits_equal(host, g["depth"])
    assert ctx.lib.mod_depth_image_host(ctx.h, None, host.ctypes.data) == capi.MOD_SKIP_NO_DISPARITY_NOW
    ctx.close()


def test_configuration_errors():
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.pipeline import Context
    ctx = Context(64, 48, max_frames=1)
    ws = ctx.workspace(1)
    z = torch.zeros((1, 48, 64), device=ctx.device)
    b = ctx.make_batch(z, z, torch.zeros((1, 48, 64, 2), device=ctx.device), [[0, 0, 0]], [[0, 0, 0, 1]], [0.1])
    with pytest.raises(capi.ModError) as e:
        ctx.process(b, ws)                                 # camera / params not set
    assert e.value.code == capi.MOD_ERR_NOT_CONFIGURED
    with pytest.raises(capi.ModError):
        ctx.set_camera(synth.make_camera(128, 48))        # larger than the context was created for
    with pytest.raises(capi.ModError):
        ctx.set_params(synth.Params(neighbor_distance=17))
    with pytest.raises(capi.ModError):
        ctx.set_params(synth.Params(cluster_size=0))
    with pytest.raises(capi.ModError) as e:
        ctx.set_params(synth.Params(cluster_size=10))      # 64*48/10 clusters could survive, capacity is 64*48/100
    assert e.value.code == capi.MOD_ERR_CAPACITY
    ctx.close()
