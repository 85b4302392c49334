"""GPU: the C-ABI entry points no other test drives directly — mod_pack_cloud_dev / mod_unpack_cloud_dev, the device-memory
helpers a non-HIP host language uses to own HBM buffers (mod_malloc / mod_free / mod_memcpy_h2d / mod_memcpy_d2h), and
mod_forget_previous."""
import ctypes as C

import numpy as np
import pytest

from util import PLANES, bits_equal

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_pack_and_unpack_cloud_dev(oracle):
    """pcl::toROSMsg / fromROSMsg payloads (scene_flow_constructor.cpp:358-361, clusterer_nodelet.cpp:226): packing the planes
    gives the records the fused kernel writes itself (and the oracle's cloud); unpacking gives the planes back."""
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.pipeline import Context
    W, H, F = 322, 97, 2                                   # even, not a multiple of 4
    cam, batch = synth.make_batch(W, H, F, seed=12)
    prm = synth.Params(dynamic_flow_diff=1, cluster_size=100)
    ctx = Context(W, H, max_frames=F)
    ctx.set_camera(cam)
    ctx.set_params(prm)
    ws = ctx.workspace(F, aos=True)
    dev = ctx.device
    b = ctx.make_batch(*(torch.from_numpy(batch[k]).to(dev) for k in ("disparity_now", "disparity_prev", "flow")),
                       batch["t"], batch["q"], batch["dt"])
    assert ctx.scene_flow(b, ws) == 0
    packed = torch.full((F, H, W, 8), -3.0, dtype=torch.float32, device=dev)
    pl = ctx._planes_struct(ws)
    assert ctx.lib.mod_pack_cloud_dev(ctx.h, F, C.byref(pl), packed.data_ptr()) == 0
    ctx.synchronize()
    got, fused = packed.cpu().numpy(), ws["aos"].cpu().numpy()
    assert bits_equal(got[..., [0, 1, 2, 4, 5, 6]], fused[..., [0, 1, 2, 4, 5, 6]])
    assert (got[..., 3] == 0).all() and (got[..., 7] == 0).all()                     # the two pad floats are written as 0
    for f in range(F):
        ref = oracle.construct(cam, prm, batch["disparity_now"][f], batch["disparity_prev"][f], batch["flow"][f], batch["t"][f],
                               batch["q"][f], float(batch["dt"][f]), "faithful")
        for j, k in zip((0, 1, 2, 4, 5, 6), PLANES):
            assert bits_equal(got[f, ..., j], ref["cloud"][k]), (f, k)
    # unpack into fresh planes
    ws2 = {"planes": torch.full((6, F, H, W), -3.0, dtype=torch.float32, device=dev), "mask": ws["mask"]}
    pl2 = ctx._planes_struct(ws2)
    assert ctx.lib.mod_unpack_cloud_dev(ctx.h, F, packed.data_ptr(), C.byref(pl2)) == 0
    ctx.synchronize()
    assert bits_equal(ws2["planes"].cpu().numpy(), ws["planes"].cpu().numpy())
    # argument checks
    assert ctx.lib.mod_pack_cloud_dev(ctx.h, F, C.byref(pl), None) == capi.MOD_ERR_INVALID_ARGUMENT
    assert ctx.lib.mod_pack_cloud_dev(ctx.h, F + 1, C.byref(pl), packed.data_ptr()) == capi.MOD_ERR_CAPACITY
    ctx.close()


def test_device_memory_helpers_carry_a_whole_frame(oracle):
    """A host language without HIP owns its HBM buffers through mod_malloc / mod_memcpy_*: one frame pair processed with no
    torch tensor anywhere on the data path, compared with the oracle."""
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.pipeline import OBJECT_DTYPE, Context
    W, H = 160, 120
    cam, batch = synth.make_batch(W, H, 1, seed=22)
    prm = synth.Params(dynamic_flow_diff=1, cluster_size=60)
    ctx = Context(W, H, max_frames=1, max_objects=W * H // 60 + 1, use_torch_stream=False)   # the context's own stream
    ctx.set_camera(cam)
    ctx.set_params(prm)
    L, h, N = ctx.lib, ctx.h, W * H

    def dmalloc(nbytes):
        p = C.c_void_p()
        assert L.mod_malloc(h, nbytes, C.byref(p)) == 0 and p.value
        return p

    d_now, d_prev, d_flow = dmalloc(4 * N), dmalloc(4 * N), dmalloc(8 * N)
    planes = [dmalloc(4 * N) for _ in range(6)]
    d_lab, d_obj, d_n = dmalloc(4 * N), dmalloc(capi.MOD_OBJECT_BYTES * ctx.max_objects), dmalloc(8)
    for dst, src in ((d_now, batch["disparity_now"][0]), (d_prev, batch["disparity_prev"][0]), (d_flow, batch["flow"][0])):
        src = np.ascontiguousarray(src)
        assert L.mod_memcpy_h2d(h, dst, src.ctypes.data, src.nbytes) == 0
    b = capi.ModFrameBatch()
    b.frames, b.disparity_now, b.disparity_prev, b.flow = 1, d_now.value, d_prev.value, d_flow.value
    b.transforms = capi.transforms_array(batch["t"], batch["q"])
    b.dt = (C.c_double * 1)(float(batch["dt"][0]))
    pl = capi.ModSceneFlowPlanes(*[p.value for p in planes], None, None, None, None)
    out = capi.ModClusterOut(d_lab.value, d_obj.value, d_n.value, d_n.value + 4)
    assert L.mod_process_dev(h, C.byref(b), C.byref(pl), C.byref(out)) == 0
    ref = oracle.construct(cam, prm, batch["disparity_now"][0], batch["disparity_prev"][0], batch["flow"][0], batch["t"][0],
                           batch["q"][0], float(batch["dt"][0]), "tidy")
    for p, k in zip(planes, PLANES):
        got = np.empty((H, W), np.float32)
        assert L.mod_memcpy_d2h(h, got.ctypes.data, p, got.nbytes) == 0            # synchronises the context's stream
        assert bits_equal(got, ref[k]), k
    labels, objs, K = oracle.cluster(ref, prm, "tidy")
    got_lab = np.empty((H, W), np.int32)
    cnt = np.zeros(2, np.int32)
    assert L.mod_memcpy_d2h(h, got_lab.ctypes.data, d_lab, got_lab.nbytes) == 0
    assert L.mod_memcpy_d2h(h, cnt.ctypes.data, d_n, 8) == 0
    assert np.array_equal(got_lab, labels) and cnt[0] == len(objs) and cnt[1] == K and K > 0
    got_obj = np.zeros(cnt[0], OBJECT_DTYPE)
    assert L.mod_memcpy_d2h(h, got_obj.ctypes.data, d_obj, got_obj.nbytes) == 0
    assert [int(o["n_points"]) for o in got_obj] == [o["n_points"] for o in objs]
    for p in [d_now, d_prev, d_flow, d_lab, d_obj, d_n] + planes:
        assert L.mod_free(h, p) == 0
    assert L.mod_malloc(None, 16, C.byref(C.c_void_p())) == capi.MOD_ERR_INVALID_ARGUMENT
    ctx.close()


def test_forget_previous():
    """disparity_now_.reset() after a failed estimateDisparity (scene_flow_constructor.cpp:272-276): the next frame must not pair
    with the stale resident disparity."""
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.pipeline import Context
    W, H = 64, 48
    cam, b = synth.make_batch(W, H, 2, seed=3)
    ctx = Context(W, H, max_frames=1)
    ctx.set_camera(cam)
    ctx.set_params(synth.Params())
    tfs = capi.transforms_array(b["t"], b["q"])
    t, n = C.c_int32(-1), C.c_int32(0)
    dn, dp, fl = (b[k][0].ctypes.data for k in ("disparity_now", "disparity_prev", "flow"))
    sub = lambda a0, a1: ctx.lib.mod_submit_frame_host(ctx.h, a0, a1, fl, C.byref(tfs[0]), 0.1, None, None, None, 0, C.byref(t))
    assert sub(dn, dp) == 0
    assert ctx.lib.mod_collect_frame_host(ctx.h, t.value, C.byref(n)) == 0
    assert sub(dn, None) == 0                                  # the resident disparity of the frame before serves as "previous"
    assert ctx.lib.mod_collect_frame_host(ctx.h, t.value, C.byref(n)) == 0
    assert ctx.lib.mod_forget_previous(ctx.h) == 0
    assert sub(dn, None) == capi.MOD_SKIP_NO_DISPARITY_PREV and t.value == -1
    assert sub(dn, dp) == 0                                    # an explicit previous disparity still works, and re-arms the chain
    assert ctx.lib.mod_collect_frame_host(ctx.h, t.value, C.byref(n)) == 0
    assert sub(dn, None) == 0
    assert ctx.lib.mod_collect_frame_host(ctx.h, t.value, C.byref(n)) == 0
    assert ctx.lib.mod_forget_previous(None) == capi.MOD_ERR_INVALID_ARGUMENT
    ctx.close()
