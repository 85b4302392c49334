"""GPU: mod_submit_frame_host / mod_collect_frame_host (frames in flight, copies on their own streams) against the
synchronous mod_process_frame_host on the same frames — bit-identical clouds, labels and objects; skip codes and the
resident previous disparity behave like construct()'s guards and the reference's disparity_previous_."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _sequence(W, H, F, seed):
    """A real sequence: frame f's previous disparity IS frame f-1's disparity."""
    from moving_object_detector_amd import synth
    cam, b = synth.make_batch(W, H, F, seed=seed)
    for f in range(1, F):
        b["disparity_prev"][f] = b["disparity_now"][f - 1]
    return cam, b


def test_stream_matches_synchronous_path():
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.pipeline import Context
    W, H, F, CAP = 320, 240, 7, 32
    cam, b = _sequence(W, H, F, seed=21)
    prm = synth.Params(cluster_size=150)
    N = W * H
    tfs = capi.transforms_array(b["t"], b["q"])

    def run(stream):
        ctx = Context(W, H, max_frames=1)
        ctx.set_camera(cam)
        ctx.set_params(prm)
        clouds = np.zeros((F, N, 8), np.float32)
        labels = np.full((F, N), -7, np.int32)
        objs = [(capi.ModObject * CAP)() for _ in range(F)]
        counts = []
        n = C.c_int32(-1)
        if not stream:
            for f in range(F):
                rc = ctx.lib.mod_process_frame_host(ctx.h, b["disparity_now"][f].ctypes.data, b["disparity_prev"][f].ctypes.data,
                                                    b["flow"][f].ctypes.data, C.byref(tfs[f]), float(b["dt"][f]), clouds[f].ctypes.data,
                                                    labels[f].ctypes.data, objs[f], CAP, C.byref(n))
                assert rc == 0
                counts.append(n.value)
        else:
            tickets = []
            t = C.c_int32(-1)
            for f in range(F):
                if len(tickets) == capi.MOD_PIPELINE_DEPTH:            # full: one more submit is refused, nothing is lost
                    rc = ctx.lib.mod_submit_frame_host(ctx.h, b["disparity_now"][f].ctypes.data, None, b["flow"][f].ctypes.data,
                                                       C.byref(tfs[f]), float(b["dt"][f]), None, None, None, 0, C.byref(t))
                    assert rc == capi.MOD_ERR_CAPACITY and t.value == -1
                    assert ctx.lib.mod_collect_frame_host(ctx.h, tickets.pop(0), C.byref(n)) == 0
                    counts.append(n.value)
                # frame 0 brings its previous disparity; later frames rely on the resident one (odd frames pass it anyway)
                dp = b["disparity_prev"][f].ctypes.data if (f == 0 or f % 2 == 1) else None
                rc = ctx.lib.mod_submit_frame_host(ctx.h, b["disparity_now"][f].ctypes.data, dp, b["flow"][f].ctypes.data, C.byref(tfs[f]),
                                                   float(b["dt"][f]), clouds[f].ctypes.data, labels[f].ctypes.data, objs[f], CAP, C.byref(t))
                assert rc == 0, ctx.lib.mod_last_error(ctx.h)
                tickets.append(t.value)
            # out-of-order collection is refused
            if len(tickets) > 1:
                assert ctx.lib.mod_collect_frame_host(ctx.h, tickets[-1], C.byref(n)) == capi.MOD_ERR_INVALID_ARGUMENT
            while tickets:
                assert ctx.lib.mod_collect_frame_host(ctx.h, tickets.pop(0), C.byref(n)) == 0
                counts.append(n.value)
            assert ctx.lib.mod_collect_frame_host(ctx.h, 0, C.byref(n)) == capi.MOD_ERR_INVALID_ARGUMENT   # nothing in flight
        ctx.close()
        return clouds, labels, objs, counts

    c0, l0, o0, n0 = run(False)
    c1, l1, o1, n1 = run(True)
    assert n0 == n1 and max(n0) > 0
    assert np.array_equal(l0, l1)
    assert np.array_equal(c0.view(np.uint32)[..., [0, 1, 2, 4, 5, 6]], c1.view(np.uint32)[..., [0, 1, 2, 4, 5, 6]])
    for f in range(F):
        a = np.frombuffer(bytes(o0[f]), np.uint8)[: 112 * n0[f]]
        bb = np.frombuffer(bytes(o1[f]), np.uint8)[: 112 * n1[f]]
        assert np.array_equal(a, bb), f


def test_stream_skip_codes_and_resident_previous_disparity():
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.pipeline import Context
    W, H = 64, 48
    cam, b = _sequence(W, H, 2, seed=3)
    ctx = Context(W, H, max_frames=1)
    ctx.set_camera(cam)
    ctx.set_params(synth.Params())
    tfs = capi.transforms_array(b["t"], b["q"])
    t, n = C.c_int32(5), C.c_int32(0)
    dn, dp, fl = (b[k][0].ctypes.data for k in ("disparity_now", "disparity_prev", "flow"))
    sub = lambda a0, a1, a2, tf: ctx.lib.mod_submit_frame_host(ctx.h, a0, a1, a2, tf, 0.1, None, None, None, 0, C.byref(t))
    assert sub(dn, dp, None, C.byref(tfs[0])) == capi.MOD_SKIP_NO_FLOW and t.value == -1
    assert sub(dn, None, fl, C.byref(tfs[0])) == capi.MOD_SKIP_NO_DISPARITY_PREV        # first frame: nothing resident yet
    assert sub(dn, dp, fl, None) == capi.MOD_SKIP_NO_TRANSFORM
    assert sub(None, dp, fl, C.byref(tfs[0])) == capi.MOD_SKIP_NO_DISPARITY_NOW
    assert sub(dn, dp, fl, C.byref(tfs[0])) == 0 and t.value == 0
    assert sub(b["disparity_now"][1].ctypes.data, None, b["flow"][1].ctypes.data, C.byref(tfs[1])) == 0 and t.value == 1
    assert ctx.lib.mod_collect_frame_host(ctx.h, 0, C.byref(n)) == 0
    assert ctx.lib.mod_collect_frame_host(ctx.h, 1, C.byref(n)) == 0
    ctx.close()


def test_pinned_host_memory_helpers():
    from moving_object_detector_amd.pipeline import Context
    ctx = Context(64, 48, max_frames=1)
    p = C.c_void_p()
    assert ctx.lib.mod_host_malloc(ctx.h, 1 << 20, C.byref(p)) == 0 and p.value
    C.memset(p.value, 0x5A, 1 << 20)
    assert ctx.lib.mod_host_free(ctx.h, p) == 0
    ctx.close()


def test_stereo_stream_matches_estimate_then_process():
    """mod_submit_stereo_host (images in, disparity resident in HBM as `now` and next frame's `previous`) against the two-call path
    mod_sgm_compute_host + mod_process_frame_host, which carries the disparity through the host: same disparity, cloud, labels,
    objects, byte for byte — on the committed 320 x 240 image pair and four more frames of the same scene; guards as construct()."""
    import os
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.pipeline import Context
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sgm_320x240.npz"))
    H, W = g["left"].shape
    D, F, CAP = 128, 5, 32
    N = W * H
    lefts, rights, truths = [np.ascontiguousarray(g["left"])], [np.ascontiguousarray(g["right"])], [g["truth"]]
    for k in range(1, F):
        l, r, t = synth.make_stereo_images(W, H, 11, D)             # the fixture's scene again (seed 11): a camera that stands still
        lefts.append(l); rights.append(r); truths.append(t)
    assert np.array_equal(lefts[1], lefts[0])
    cam = synth.make_camera(W, H)
    cam.min_disparity, cam.max_disparity = np.float32(0.0), np.float32(D - 1)
    prm = synth.Params(cluster_size=150)
    flows = [synth.make_box_flow(truths[k], shift=12.0 + k) for k in range(F)]      # the boxes move, a little more every frame
    tf = capi.transforms_array(np.zeros((F, 3)), np.tile(np.array([[0.0, 0.0, 0.0, 1.0]]), (F, 1)))
    sp = capi.ModSgmParams(D, 6, 96, 8, 1, 1)
    dt = 1.0 / 15.0

    def two_calls():
        ctx = Context(W, H, max_frames=1)
        ctx.set_camera(cam); ctx.set_params(prm)
        disp = np.zeros((F, H, W), np.float32)
        clouds = np.zeros((F, N, 8), np.float32); labels = np.full((F, N), -7, np.int32)
        objs = [(capi.ModObject * CAP)() for _ in range(F)]
        counts, n = [], C.c_int32(-1)
        for f in range(F):
            assert ctx.lib.mod_sgm_compute_host(ctx.h, lefts[f].ctypes.data, rights[f].ctypes.data, C.byref(sp), disp[f].ctypes.data) == 0
            if f == 0:
                counts.append(None)                                   # no previous disparity: construct() publishes nothing
                continue
            rc = ctx.lib.mod_process_frame_host(ctx.h, disp[f].ctypes.data, disp[f - 1].ctypes.data, flows[f].ctypes.data, C.byref(tf[f]), dt,
                                                clouds[f].ctypes.data, labels[f].ctypes.data, objs[f], CAP, C.byref(n))
            assert rc == 0
            counts.append(n.value)
        ctx.close()
        return disp, clouds, labels, objs, counts

    def one_call():
        ctx = Context(W, H, max_frames=1)
        ctx.set_camera(cam); ctx.set_params(prm)
        disp = np.zeros((F, H, W), np.float32)
        clouds = np.zeros((F, N, 8), np.float32); labels = np.full((F, N), -7, np.int32)
        objs = [(capi.ModObject * CAP)() for _ in range(F)]
        counts, tickets, t, n = [], [], C.c_int32(-1), C.c_int32(-1)
        sub = lambda f, fl, tfp, d=None: ctx.lib.mod_submit_stereo_host(
            ctx.h, lefts[f].ctypes.data, rights[f].ctypes.data, C.byref(sp), fl, tfp, dt, clouds[f].ctypes.data, labels[f].ctypes.data, objs[f], CAP,
            d, C.byref(t))
        # frame 0: no previous disparity yet -> skip code, but its disparity stays as frame 1's previous one
        assert sub(0, flows[0].ctypes.data, C.byref(tf[0])) == capi.MOD_SKIP_NO_DISPARITY_PREV and t.value == -1
        counts.append(None)
        for f in range(1, F):
            if f == 3:                                                # a frame without flow in between: skipped, yet it is frame 4's previous
                assert sub(f, None, C.byref(tf[f])) == capi.MOD_SKIP_NO_FLOW and t.value == -1
                counts.append(None)
                continue
            assert sub(f, flows[f].ctypes.data, C.byref(tf[f]), disp[f].ctypes.data) == 0, ctx.lib.mod_last_error(ctx.h)
            tickets.append(t.value)
        for tk in tickets:
            assert ctx.lib.mod_collect_frame_host(ctx.h, tk, C.byref(n)) == 0
            counts.append(n.value)
        # no images: the estimator has nothing, and the NEXT frame has no previous disparity
        assert ctx.lib.mod_submit_stereo_host(ctx.h, None, rights[0].ctypes.data, C.byref(sp), flows[0].ctypes.data, C.byref(tf[0]), dt, None, None, None, 0,
                                              None, C.byref(t)) == capi.MOD_SKIP_NO_DISPARITY_NOW
        assert sub(1, flows[1].ctypes.data, C.byref(tf[1])) == capi.MOD_SKIP_NO_DISPARITY_PREV
        ctx.close()
        return disp, clouds, labels, objs, counts

    d0, c0, l0, o0, n0 = two_calls()
    d1, c1, l1, o1, n1 = one_call()
    assert np.array_equal(d0[0], g["disparity"])                       # the fixture pins the estimator
    got = [n for n in n1 if n is not None]
    want = [n0[f] for f in (1, 2, 4)]
    assert got == want and max(want) > 0, (got, want)
    for f in (1, 2, 4):
        assert np.array_equal(d1[f], d0[f]), f                          # optional host copy of the device-resident plane
        assert np.array_equal(l1[f], l0[f]), f
        assert np.array_equal(c1[f].view(np.uint32)[..., [0, 1, 2, 4, 5, 6]], c0[f].view(np.uint32)[..., [0, 1, 2, 4, 5, 6]]), f
        k = n0[f]
        assert np.array_equal(np.frombuffer(bytes(o1[f]), np.uint8)[: 112 * k], np.frombuffer(bytes(o0[f]), np.uint8)[: 112 * k]), f


def test_stereo_and_disparity_submissions_share_the_ring():
    """The two streaming entries may alternate on one context: a frame submitted as IMAGES leaves its disparity in the pipe's ring, and a
    following frame submitted as a DISPARITY (mod_submit_frame_host with disparity_prev = NULL) pairs with it — also when the image
    frame ended at a guard and took no ticket (the host copy then queues behind the estimator that is still writing the plane)."""
    import os
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.pipeline import Context
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sgm_320x240.npz"))
    H, W = g["left"].shape
    D, N, CAP = 128, H * W, 32
    left, right = np.ascontiguousarray(g["left"]), np.ascontiguousarray(g["right"])
    cam = synth.make_camera(W, H)
    cam.min_disparity, cam.max_disparity = np.float32(0.0), np.float32(D - 1)
    prm = synth.Params(cluster_size=150)
    sp = capi.ModSgmParams(D, 6, 96, 8, 1, 1)
    flow = synth.make_box_flow(g["truth"], shift=14.0)
    tf = capi.transforms_array(np.zeros((1, 3)), np.array([[0.0, 0.0, 0.0, 1.0]]))
    dt = 1.0 / 15.0
    disp = np.ascontiguousarray(g["disparity"])                        # the estimator's result for this pair (pinned by the fixture)

    ctx = Context(W, H, max_frames=1)
    ctx.set_camera(cam); ctx.set_params(prm)
    t, n = C.c_int32(-1), C.c_int32(-1)
    # frame 0 as images, no flow: ends at a guard, no ticket — but its disparity is now the resident previous plane
    rc = ctx.lib.mod_submit_stereo_host(ctx.h, left.ctypes.data, right.ctypes.data, C.byref(sp), None, C.byref(tf[0]), dt, None, None, None, 0, None, C.byref(t))
    assert rc == capi.MOD_SKIP_NO_FLOW and t.value == -1
    # frame 1 as a disparity image, previous = the plane the estimator left behind
    cloud1 = np.zeros((N, 8), np.float32); lab1 = np.full(N, -7, np.int32); objs1 = (capi.ModObject * CAP)()
    rc = ctx.lib.mod_submit_frame_host(ctx.h, disp.ctypes.data, None, flow.ctypes.data, C.byref(tf[0]), dt, cloud1.ctypes.data, lab1.ctypes.data, objs1, CAP, C.byref(t))
    assert rc == 0, ctx.lib.mod_last_error(ctx.h)
    assert ctx.lib.mod_collect_frame_host(ctx.h, t.value, C.byref(n)) == 0
    n1 = n.value
    ctx.close()

    ref = Context(W, H, max_frames=1)
    ref.set_camera(cam); ref.set_params(prm)
    cloud0 = np.zeros((N, 8), np.float32); lab0 = np.full(N, -7, np.int32); objs0 = (capi.ModObject * CAP)()
    rc = ref.lib.mod_process_frame_host(ref.h, disp.ctypes.data, disp.ctypes.data, flow.ctypes.data, C.byref(tf[0]), dt, cloud0.ctypes.data, lab0.ctypes.data, objs0, CAP, C.byref(n))
    assert rc == 0 and n.value == n1 and n1 > 0
    ref.close()
    assert np.array_equal(lab1, lab0)
    assert np.array_equal(cloud1.view(np.uint32)[:, [0, 1, 2, 4, 5, 6]], cloud0.view(np.uint32)[:, [0, 1, 2, 4, 5, 6]])
    assert bytes(objs1)[: 112 * n1] == bytes(objs0)[: 112 * n1]


def test_disparity_copy_survives_guard_skipped_frames_that_wrap_the_ring():
    """A ticketed stereo frame asks for its disparity plane on the host; before it is collected, MOD_PIPELINE_DEPTH + 1 frames that end
    at a guard (no flow: a ring plane each, no ticket) bring the ring back to that plane, and the last one's estimator overwrites it
    with ANOTHER image pair's disparity.  The plane's writer has to wait for the copy: the host buffer must hold the first pair's
    disparity."""
    import os
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.pipeline import Context
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sgm_320x240.npz"))
    H, W = g["left"].shape
    D, N, CAP = 128, H * W, 32
    left, right = np.ascontiguousarray(g["left"]), np.ascontiguousarray(g["right"])
    other_l, other_r, _ = synth.make_stereo_images(W, H, 5, D)          # a different scene
    cam = synth.make_camera(W, H)
    cam.min_disparity, cam.max_disparity = np.float32(0.0), np.float32(D - 1)
    sp = capi.ModSgmParams(D, 6, 96, 8, 1, 1)
    flow = synth.make_box_flow(g["truth"], shift=14.0)
    tf = capi.transforms_array(np.zeros((1, 3)), np.array([[0.0, 0.0, 0.0, 1.0]]))
    dt = 1.0 / 15.0
    ctx = Context(W, H, max_frames=1)
    ctx.set_camera(cam); ctx.set_params(synth.Params(cluster_size=150))
    t, n = C.c_int32(-1), C.c_int32(-1)
    sub = lambda l, r, fl, d: ctx.lib.mod_submit_stereo_host(ctx.h, l.ctypes.data, r.ctypes.data, C.byref(sp), fl, C.byref(tf[0]), dt, None, None, None, 0, d, C.byref(t))
    assert sub(left, right, None, None) == capi.MOD_SKIP_NO_FLOW         # gives the next frame its previous disparity
    disp = torch.zeros((H, W), dtype=torch.float32).pin_memory().numpy() # pinned: the copy really is asynchronous
    assert sub(left, right, flow.ctypes.data, disp.ctypes.data) == 0, ctx.lib.mod_last_error(ctx.h)
    ticket = t.value
    for k in range(capi.MOD_PIPELINE_DEPTH + 1):                         # the ring has DEPTH + 1 planes: the last of these is the ticketed frame's
        assert sub(other_l, other_r, None, None) == capi.MOD_SKIP_NO_FLOW
    assert ctx.lib.mod_collect_frame_host(ctx.h, ticket, C.byref(n)) == 0
    ctx.synchronize()
    ctx.close()
    assert np.array_equal(disp, g["disparity"])


def test_parameters_may_change_between_submits(oracle):
    """include/mod_sf.h: the PARAMETERS may be reconfigured while frames are in flight — the reference's reconfigureCB runs between two
    stereoCallbacks (scene_flow_constructor.cpp:401-407, clusterer_nodelet.cpp:345-352), i.e. after frame t was handed to
    construct_thread_ and before it is joined — and every kernel takes them by value at submit time: a frame in flight completes with
    the parameters of ITS submit, the next submit uses the new ones.  The second change raises neighbor_distance, which re-allocates
    the link-request scratch (mod_set_params synchronises the stream for that)."""
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.pipeline import Context, OBJECT_DTYPE
    from util import compare_objects
    W, H, F, CAP = 320, 240, 4, 64
    cam, b = _sequence(W, H, F, seed=33)
    prms = [synth.Params(dynamic_flow_diff=1, cluster_size=100, neighbor_distance=2),
            synth.Params(dynamic_flow_diff=1, cluster_size=400, neighbor_distance=2, depth_diff=0.05),
            synth.Params(dynamic_flow_diff=2, cluster_size=100, neighbor_distance=9, dynamic_speed=0.2),
            synth.Params(dynamic_flow_diff=1, cluster_size=150, neighbor_distance=4)]
    tfs = capi.transforms_array(b["t"], b["q"])
    ctx = Context(W, H, max_frames=1)
    ctx.set_camera(cam)
    labels = np.full((F, H, W), -7, np.int32)
    objs = [np.zeros(CAP, OBJECT_DTYPE) for _ in range(F)]
    t, n = C.c_int32(-1), C.c_int32(-1)
    tickets, counts = [], []
    for f in range(F):
        ctx.set_params(prms[f])                              # between two submits, frames f-1, f-2 still in flight
        if len(tickets) == capi.MOD_PIPELINE_DEPTH:
            assert ctx.lib.mod_collect_frame_host(ctx.h, tickets.pop(0), C.byref(n)) == 0
            counts.append(n.value)
        rc = ctx.lib.mod_submit_frame_host(ctx.h, b["disparity_now"][f].ctypes.data, b["disparity_prev"][f].ctypes.data, b["flow"][f].ctypes.data,
                                           C.byref(tfs[f]), float(b["dt"][f]), None, labels[f].ctypes.data, objs[f].ctypes.data, CAP, C.byref(t))
        assert rc == 0, ctx.lib.mod_last_error(ctx.h)
        tickets.append(t.value)
    ctx.set_params(synth.Params(cluster_size=5000))          # after the last submit, before its collect: must not reach back
    while tickets:
        assert ctx.lib.mod_collect_frame_host(ctx.h, tickets.pop(0), C.byref(n)) == 0
        counts.append(n.value)
    ctx.close()
    total = 0
    for f in range(F):
        ref = oracle.construct(cam, prms[f], b["disparity_now"][f], b["disparity_prev"][f], b["flow"][f], b["t"][f], b["q"][f], float(b["dt"][f]), "tidy")
        rl, ro, _ = oracle.cluster(ref, prms[f], "tidy", max_objects=W * H)
        assert np.array_equal(labels[f], rl), f
        assert counts[f] == len(ro), (f, counts[f], len(ro))
        compare_objects(objs[f][: counts[f]], ro, strict_velocity=True)
        total += len(ro)
    assert total > 0
