"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Bar (BASELINE.json north_star): cluster labels bit-exact, 3D scene flow within 1e-4 relative — the implementation is
in fact bit-exact on every plane, and the tests assert that (tolerance written here: 0 ulp, NaN == NaN)."""
import numpy as np
import pytest

from util import PLANES, bits_equal, compare_objects, first_mismatch, rel_err

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _run_gpu(cam, prm, batch, aos=False, extras=False, fused=True):
    from moving_object_detector_amd.pipeline import Context
    F, H, W = batch["disparity_now"].shape
    ctx = Context(W, H, max_frames=F, max_objects=max(W * H // 100, W * H // prm.cluster_size + 1))
    ctx.set_camera(cam)
    ctx.set_params(prm)
    ws = ctx.workspace(F, aos=aos, extras=extras)
    dev = ctx.device
    dn = torch.from_numpy(batch["disparity_now"]).to(dev)
    dp = torch.from_numpy(batch["disparity_prev"]).to(dev)
    fl = torch.from_numpy(batch["flow"]).to(dev)
    b = ctx.make_batch(dn, dp, fl, batch["t"], batch["q"], batch["dt"])
    if fused:
        assert ctx.process(b, ws) == 0
    else:
        assert ctx.scene_flow(b, ws) == 0
        assert ctx.cluster(F, ws, mask_ready=False) == 0
    ctx.synchronize()
    out = {k: ws["planes"][i].cpu().numpy() for i, k in enumerate(PLANES)}
    out["labels"] = ws["labels"].cpu().numpy()
    out["objects"] = ctx.objects_to_host(ws)
    out["n_clusters"] = ws["n_clusters"].cpu().numpy()
    out["mask"] = ws["mask"].cpu().numpy()
    if aos:
        out["aos"] = ws["aos"].cpu().numpy()
    if extras:
        out["depth"] = ws["depth"].cpu().numpy()
        out["static_flow"] = ws["static_flow"].cpu().numpy()
    ctx.close()
    return out


def _check_against_oracle(oracle, cam, prm, batch, out, strict_velocity=True):
    F = batch["disparity_now"].shape[0]
    for f in range(F):
        ref = oracle.construct(cam, prm, batch["disparity_now"][f], batch["disparity_prev"][f], batch["flow"][f],
                               batch["t"][f], batch["q"][f], float(batch["dt"][f]), "tidy")
        for k in PLANES:
            assert bits_equal(out[k][f], ref[k]), (f, k, first_mismatch(out[k][f], ref[k]))
            assert rel_err(out[k][f], ref[k]) <= 1e-4          # the north_star tolerance, implied by bit-exactness
        labels, objs, K = oracle.cluster(ref, prm, "tidy")
        assert np.array_equal(out["labels"][f], labels), (f, int((out["labels"][f] != labels).sum()))
        assert int(out["n_clusters"][f]) == K
        compare_objects(out["objects"][f], objs, strict_velocity=strict_velocity)
    return True


@pytest.mark.parametrize("W,H,flow_diff,csize", [(64, 48, 1, 20), (160, 120, 1, 60), (333, 187, 2, 150), (640, 480, 5, 600)])
def test_fused_matches_oracle(oracle, W, H, flow_diff, csize):
    from moving_object_detector_amd import synth
    cam, batch = synth.make_batch(W, H, 2, seed=11)
    prm = synth.Params(dynamic_flow_diff=flow_diff, cluster_size=csize)
    out = _run_gpu(cam, prm, batch)
    _check_against_oracle(oracle, cam, prm, batch, out)


def test_1280x720_reference_defaults(oracle):
    from moving_object_detector_amd import synth
    cam, batch = synth.make_batch(1280, 720, 2, seed=0)
    prm = synth.Params()
    out = _run_gpu(cam, prm, batch, aos=True, extras=True)
    _check_against_oracle(oracle, cam, prm, batch, out)
    # reference-layout outputs: 32-byte PointXYZVelocity records, depth image, synthetic optical flow
    for f in range(2):
        ref = oracle.construct(cam, prm, batch["disparity_now"][f], batch["disparity_prev"][f], batch["flow"][f],
                               batch["t"][f], batch["q"][f], float(batch["dt"][f]), "faithful")
        aos = out["aos"][f]
        for j, k in zip((0, 1, 2, 4, 5, 6), PLANES):
            assert bits_equal(aos[..., j], ref[k])
        assert bits_equal(out["static_flow"][f], ref["static_flow"])
        assert bits_equal(out["depth"][f], ref["depth"]), first_mismatch(out["depth"][f], ref["depth"])   # ~depth (toDepthImage)


@pytest.mark.parametrize("n", [1, 4, 10])
def test_neighbor_distance_sweep(oracle, n):
    from moving_object_detector_amd import synth
    cam, batch = synth.make_batch(320, 240, 1, seed=5)
    prm = synth.Params(dynamic_flow_diff=1, cluster_size=100, neighbor_distance=n)
    out = _run_gpu(cam, prm, batch)
    _check_against_oracle(oracle, cam, prm, batch, out)


def test_unfused_equals_fused(oracle):
    from moving_object_detector_amd import synth
    cam, batch = synth.make_batch(320, 240, 2, seed=9)
    prm = synth.Params(dynamic_flow_diff=1, cluster_size=100)
    a = _run_gpu(cam, prm, batch, fused=True)
    b = _run_gpu(cam, prm, batch, fused=False)
    assert np.array_equal(a["labels"], b["labels"])
    assert np.array_equal(a["mask"], b["mask"])
    for k in PLANES:
        assert bits_equal(a[k], b[k])


def test_dynamic_mask_is_exact_at_the_threshold():
    """calculateDynamicMap compares (double)||v||_f32 >= dynamic_speed; ||v|| needs an IEEE square root (HIP's __fsqrt_rn
    is the *native* approximation) and the F64 threshold is folded into an F32 one on the host.  Thousands of norms sit
    within a few ulps of the threshold here, including exact ties."""
    import ctypes as C
    from moving_object_detector_amd import synth
    from moving_object_detector_amd.pipeline import Context
    from oracle import numpy_ref
    W, H = 512, 64
    rng = np.random.default_rng(77)
    d = rng.normal(size=(3, H, W))
    d /= np.linalg.norm(d, axis=0)
    scale = 0.7 * (1.0 + rng.integers(-40, 41, size=(H, W)) * 2.0 ** -23)      # norms within ~40 ulp of 0.7
    v = (d * scale).astype(np.float32)
    v[:, 0, :8] = np.nan
    v[:, 1, :8] = 0.0
    for th in (float(numpy_ref.norm3(v[0], v[1], v[2])[5, 5]), 0.7, float(np.float32(0.7)), np.nextafter(0.7, 1.0)):
        prm = synth.Params(dynamic_speed=th)
        ctx = Context(W, H, max_frames=1)
        ctx.set_camera(synth.make_camera(W, H))
        ctx.set_params(prm)
        t = [torch.from_numpy(v[i][None].copy()).to(ctx.device) for i in range(3)]
        mask = torch.zeros((1, H, W // 64), dtype=torch.int64, device=ctx.device)
        assert ctx.lib.mod_dynamic_mask_dev(ctx.h, 1, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), mask.data_ptr()) == 0
        ctx.synchronize()
        bits = np.unpackbits(mask.cpu().numpy().view(np.uint8), axis=-1, bitorder="little")[0].astype(bool)
        want = numpy_ref.dynamic_mask(prm, v[0], v[1], v[2])
        assert np.array_equal(bits, want), (th, int((bits != want).sum()))
        assert 0.02 < want.mean() < 0.98
        ctx.close()


def test_kitti_style_camera(oracle):
    """BASELINE.json config 3's data shape: KITTI-style intrinsics / baseline, dt = 0.1 s."""
    from moving_object_detector_amd import synth
    cam, batch = synth.make_batch(496, 150, 2, seed=17, camera="kitti")
    assert abs(cam.fx - 721.5377 * 496 / 1242) < 1e-9 and float(cam.disp_T) == np.float32(0.5372)
    prm = synth.Params(cluster_size=300)
    out = _run_gpu(cam, prm, batch)
    _check_against_oracle(oracle, cam, prm, batch, out)


def test_kitti_style_camera_1280x720_sequence(oracle):
    """Config 3 at its stated size: a 1280x720 KITTI-style *sequence* (frame t's now-disparity is frame t+1's previous, one
    buffer of F+1 planes: prev = D, now = D + H*W), reference default parameters."""
    from moving_object_detector_amd import synth
    cam, seq = synth.make_sequence(1280, 720, 3, seed=31, camera="kitti")
    batch = {"disparity_now": seq["disparity"][1:], "disparity_prev": seq["disparity"][:-1], "flow": seq["flow"],
             "t": seq["t"], "q": seq["q"], "dt": seq["dt"]}
    batch = {k: np.ascontiguousarray(v) for k, v in batch.items()}
    prm = synth.Params()
    out = _run_gpu(cam, prm, batch, extras=True)
    _check_against_oracle(oracle, cam, prm, batch, out)
    assert sum(len(o) for o in out["objects"]) > 0
    for f in range(3):
        assert bits_equal(out["depth"][f], oracle.depth_image(cam, batch["disparity_now"][f]))


def test_nominal_workload_1280x720_sequence(oracle):
    """SURVEY.md section 8(d)'s NOMINAL synthetic workload (object depth 4-15 m, speed 0.5-2 m/s, dt = 1/15 s; the default stream uses
    4-9 m, 1-2 m/s, 0.1 s so that more objects pass the reference's 5 px residual test — DESIGN.md section 10): three frames at
    1280x720, reference default parameters, bit for bit against the oracle (`bench.py --workload nominal` times this stream)."""
    from moving_object_detector_amd import synth
    cam, seq = synth.make_sequence(1280, 720, 3, seed=4, **synth.NOMINAL)
    batch = {"disparity_now": seq["disparity"][1:], "disparity_prev": seq["disparity"][:-1], "flow": seq["flow"],
             "t": seq["t"], "q": seq["q"], "dt": seq["dt"]}
    batch = {k: np.ascontiguousarray(v) for k, v in batch.items()}
    assert float(batch["dt"][0]) == 1.0 / 15.0
    prm = synth.Params()
    out = _run_gpu(cam, prm, batch)
    _check_against_oracle(oracle, cam, prm, batch, out)
    assert sum(len(o) for o in out["objects"]) >= 6


def test_4k_frame(oracle):
    """Largest shape exercised: one 3840x2160 pair (8.3 M px, 9x the 720p frame: 32-bit in-frame byte offsets, 60 x 135 tiles,
    clusters of several hundred thousand pixels), reference default parameters."""
    from moving_object_detector_amd import synth
    cam, batch = synth.make_batch(3840, 2160, 1, seed=23)
    prm = synth.Params()
    out = _run_gpu(cam, prm, batch, aos=True)
    _check_against_oracle(oracle, cam, prm, batch, out)
    assert len(out["objects"][0]) >= 1
