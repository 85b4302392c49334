"""GPU: mod_process_dev on a large batch runs its cluster stage in chunks of frames on streams of the context's own
(ModConfig.batch_chunks, include/mod_sf.h) — a scheduling choice that must not show in any output — and every call leaves the
cluster scratch (tile headers, counters) as it found it, so that no call has to clear it first (mod_sf.hip: scratch_clean)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _stream(W, H, F, seed=4):
    from moving_object_detector_amd import synth
    cam, sq = synth.make_sequence(W, H, F, seed=seed)
    return cam, sq


def _ctx(W, H, F, cam, prm, chunks, **kw):
    from moving_object_detector_amd.pipeline import Context
    ctx = Context(W, H, max_frames=F, max_objects=W * H // prm.cluster_size + 1, batch_chunks=chunks, **kw)
    ctx.set_camera(cam); ctx.set_params(prm)
    return ctx


def _batch(ctx, sq, lo, hi):
    dev = ctx.device
    d = torch.from_numpy(sq["disparity"]).to(dev)
    return ctx.make_batch(d[1 + lo:1 + hi].contiguous(), d[lo:hi].contiguous(), torch.from_numpy(sq["flow"][lo:hi]).to(dev),
                          sq["t"][lo:hi], sq["q"][lo:hi], sq["dt"][lo:hi])


def _outputs(ctx, ws):
    n = ws["n_objects"].cpu().numpy().copy()
    raw = ws["objects"].cpu().numpy()
    return (ws["planes"].cpu().numpy().copy(), ws["labels"].cpu().numpy().copy(), n, ws["n_clusters"].cpu().numpy().copy(),
            [raw[f, : n[f]].tobytes() for f in range(len(n))])


def _same(a, b):
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), "planes differ"
    assert np.array_equal(a[1], b[1]), "labels differ"
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]), "object / cluster counts differ"
    assert a[4] == b[4], "objects differ"


def test_chunked_batch_equals_the_unchunked_one_and_the_oracle(oracle):
    """160 frames of 320 x 240: one chunk, the default (0), 2 / 3 / 4 chunks side by side."""
    from moving_object_detector_amd import synth
    from util import compare_objects
    W, H, F = 320, 240, 160
    cam, sq = _stream(W, H, F)
    prm = synth.Params(dynamic_flow_diff=1, cluster_size=60)
    want = None
    for chunks in (1, 0, 2, 3, 4):
        ctx = _ctx(W, H, F, cam, prm, chunks)
        ws = ctx.workspace(F)
        batch = _batch(ctx, sq, 0, F)
        for rep in range(2):                                  # the second call finds the scratch as the first one left it
            ws["planes"].fill_(-1.0); ws["labels"].fill_(-7); ws["objects"].zero_(); ws["n_objects"].fill_(-1)
            assert ctx.process(batch, ws) == 0
            ctx.synchronize()
            got = _outputs(ctx, ws)
            if want is None:
                want = got
                assert int(got[2].sum()) > F // 2                # the stream does hold objects
                objs = ctx.objects_to_host(ws)
                for f in (0, 79, 80, 159):                       # chunk borders of the 2-chunk run included
                    ref = oracle.construct(cam, prm, sq["disparity"][f + 1], sq["disparity"][f], sq["flow"][f], sq["t"][f], sq["q"][f], float(sq["dt"][f]), "tidy")
                    rl, ro, _ = oracle.cluster(ref, prm, "tidy", max_objects=W * H)
                    assert np.array_equal(got[1][f], rl), f
                    compare_objects(objs[f], ro, strict_velocity=True)
            else:
                _same(got, want)
        ctx.close()


def test_invalid_chunk_settings_are_refused():
    from moving_object_detector_amd import capi
    lib = capi.load()
    for bad in (5, -1, 0x102, 0x1000002):
        h = C.c_void_p()
        cfg = capi.ModConfig(0, 64, 48, 4, 0, bad, None)
        assert lib.mod_create(C.byref(cfg), C.byref(h)) == capi.MOD_ERR_INVALID_ARGUMENT, bad


def test_per_kernel_timers_run_one_chunk_and_the_group_timer_spans_all():
    from moving_object_detector_amd import capi, synth
    W, H, F = 320, 240, 96
    cam, sq = _stream(W, H, F)
    prm = synth.Params(dynamic_flow_diff=1, cluster_size=60)
    ctx = _ctx(W, H, F, cam, prm, 3)
    ws = ctx.workspace(F)
    batch = _batch(ctx, sq, 0, F)
    assert ctx.process(batch, ws) == 0
    ctx.synchronize()
    want = _outputs(ctx, ws)
    ctx.set_profiling(True, stages=[capi.MOD_STAGE_SCENE_FLOW, capi.MOD_STAGE_CLUSTER_GROUP])
    ctx.reset_stage_times()
    for _ in range(3):
        assert ctx.process(batch, ws) == 0
    assert ctx.stage_time(capi.MOD_STAGE_SCENE_FLOW)[1] == 3            # one scene-flow launch per call, whatever the chunks
    grp_ms, grp_n = ctx.stage_time(capi.MOD_STAGE_CLUSTER_GROUP)
    assert grp_n == 3 and grp_ms > 0.0
    assert all(ctx.stage_time(i)[1] == 0 for i in capi.MOD_PER_KERNEL_CLUSTER_STAGES)
    _same(_outputs(ctx, ws), want)
    ctx.set_profiling(True)                                             # all timers: ONE chunk, every kernel timed once per call
    ctx.reset_stage_times()
    assert ctx.process(batch, ws) == 0
    assert all(ctx.stage_time(i)[1] == 1 for i in range(capi.MOD_STAGE_COUNT))
    parts = sum(ctx.stage_time(i)[0] for i in capi.MOD_PER_KERNEL_CLUSTER_STAGES)
    assert ctx.stage_time(capi.MOD_STAGE_CLUSTER_GROUP)[0] >= 0.9 * parts   # the group spans its kernels
    _same(_outputs(ctx, ws), want)
    ctx.set_profiling(False)
    ctx.close()


def test_scratch_is_left_clean_across_calls_of_different_shape(oracle):
    """One context, calls of 7, 2, 7 frames on different data, then the clusterer alone on a caller's cloud, then a fused call again:
    every call must see tile headers and counters as a fresh context does (no memset runs between them)."""
    from moving_object_detector_amd import synth
    W, H, F = 384, 200, 7
    cam, sq = _stream(W, H, 2 * F, seed=6)
    prm = synth.Params(dynamic_flow_diff=1, cluster_size=40)
    ctx = _ctx(W, H, F, cam, prm, 0)

    def fresh(lo, hi):
        c2 = _ctx(W, H, F, cam, prm, 1)
        w2 = c2.workspace(hi - lo)
        assert c2.process(_batch(c2, sq, lo, hi), w2) == 0
        c2.synchronize()
        out = _outputs(c2, w2)
        c2.close()
        return out

    for lo, hi in ((0, 7), (9, 11), (7, 14), (3, 4), (0, 7)):
        ws = ctx.workspace(hi - lo)
        ws["labels"].fill_(-7); ws["objects"].zero_(); ws["n_objects"].fill_(-1)
        assert ctx.process(_batch(ctx, sq, lo, hi), ws) == 0
        ctx.synchronize()
        _same(_outputs(ctx, ws), fresh(lo, hi))
    # the clusterer alone (k_tile_flags writes the headers of this path) between two fused calls
    ws = ctx.workspace(7)
    assert ctx.process(_batch(ctx, sq, 0, 7), ws) == 0
    ctx.synchronize()
    want = _outputs(ctx, ws)
    ws["labels"].fill_(-7); ws["objects"].zero_(); ws["n_objects"].fill_(-1)
    assert ctx.cluster(7, ws, mask_ready=False) == 0
    ctx.synchronize()
    got = _outputs(ctx, ws)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]) and got[4] == want[4]
    ws["labels"].fill_(-7); ws["objects"].zero_(); ws["n_objects"].fill_(-1)
    assert ctx.process(_batch(ctx, sq, 7, 14), ws) == 0
    ctx.synchronize()
    _same(_outputs(ctx, ws), fresh(7, 14))
    ctx.close()


def test_size_filter_inside_the_merge_kernel_and_as_a_kernel_of_its_own_agree():
    """Batches of up to 16 frames run the size filter in k_ccl_merge's last workgroup per frame, larger ones as k_select behind it
    (cluster.hip kFusedFilterFrames): 16 and 17 frames of one stream, the fused call and the clusterer alone, against one-frame calls."""
    from moving_object_detector_amd import synth
    W, H, F = 320, 240, 17
    cam, sq = _stream(W, H, F, seed=5)
    prm = synth.Params(dynamic_flow_diff=1, cluster_size=60)
    one = _ctx(W, H, 1, cam, prm, 0)
    w1 = one.workspace(1)
    single = []
    for f in range(F):
        assert one.process(_batch(one, sq, f, f + 1), w1) == 0
        one.synchronize()
        single.append(_outputs(one, w1))
    one.close()
    assert sum(int(o[2][0]) for o in single) > F // 2
    for frames in (16, 17):
        ctx = _ctx(W, H, frames, cam, prm, 0)
        ws = ctx.workspace(frames)
        for mode in ("fused", "clusterer alone"):
            ws["labels"].fill_(-7); ws["objects"].zero_(); ws["n_objects"].fill_(-1); ws["n_clusters"].fill_(-1)
            if mode == "fused":
                assert ctx.process(_batch(ctx, sq, 0, frames), ws) == 0
            else:                                            # the planes of the fused call, clustered again as a caller's cloud
                assert ctx.cluster(frames, ws, mask_ready=False) == 0
            ctx.synchronize()
            got = _outputs(ctx, ws)
            for f in range(frames):
                assert np.array_equal(got[1][f], single[f][1][0]), (mode, frames, f)
                assert got[2][f] == single[f][2][0] and got[3][f] == single[f][3][0], (mode, frames, f)
                assert got[4][f] == single[f][4][0], (mode, frames, f)
        ctx.close()
