"""Helper PROGRAM (not a test): runs the hot path of the library named by MOD_SF_LIB — the CHECKED build, whose kernels verify
every data-derived index before using it (csrc/mod_device.h MOD_CHECK) — over the shapes that matter for those indices, compares
the results with the oracle, and prints the violation counters as JSON.  Started by tests/test_gpu_checked_build.py."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch

    from moving_object_detector_amd import synth
    from moving_object_detector_amd.pipeline import Context
    from oracle import pyoracle
    from util import PLANES, bits_equal, compare_objects

    counters = np.zeros(96, np.uint64)
    ran = []

    def run(name, cam, prm, batch, check_frames):
        F, H, W = batch["disparity_now"].shape
        ctx = Context(W, H, max_frames=F, max_objects=max(W * H // 100, W * H // prm.cluster_size + 1))
        ctx.set_camera(cam)
        ctx.set_params(prm)
        ws = ctx.workspace(F)
        dev = ctx.device
        b = ctx.make_batch(*(torch.from_numpy(np.ascontiguousarray(batch[k])).to(dev) for k in ("disparity_now", "disparity_prev", "flow")),
                           batch["t"], batch["q"], batch["dt"])
        for _ in range(2):                                   # twice: the second pass runs over scratch the first one left behind
            assert ctx.process(b, ws) == 0
            ctx.synchronize()
        out = (C.c_uint64 * 96)()
        ctx.lib.mod_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        assert ctx.lib.mod_debug_counters(ctx.h, out) == 0
        counters[:] += np.frombuffer(out, np.uint64)
        planes, labels, objs = ws["planes"].cpu().numpy(), ws["labels"].cpu().numpy(), ctx.objects_to_host(ws)
        for f in check_frames:
            ref = pyoracle.construct(cam, prm, batch["disparity_now"][f], batch["disparity_prev"][f], batch["flow"][f], batch["t"][f],
                                     batch["q"][f], float(batch["dt"][f]), "tidy")
            for i, k in enumerate(PLANES):
                assert bits_equal(planes[i, f], ref[k]), (name, f, k)
            lab, ro, K = pyoracle.cluster(ref, prm, "tidy", max_objects=W * H)
            assert np.array_equal(labels[f], lab), (name, f)
            compare_objects(objs[f], ro, strict_velocity=True)
        ctx.close()
        ran.append(name)

    # 1. the shape of the round-1 fault: 64 pairs of 1280 x 720 in one batch (tile roots in every tile row incl. the last,
    #    clusters that span hundreds of tiles, roots redirected across tiles by k_ccl_merge)
    cam, b16 = synth.make_batch(1280, 720, 16, seed=0)
    idx = [i % 16 for i in range(64)]
    run("64 pairs 1280x720", cam, synth.Params(), {k: v[idx] for k, v in b16.items()}, [0, 37, 63])
    # 2. a stream (previous = now of the frame before), window 1 and 10, tiny clusters -> many cluster records and work items
    cam, seq = synth.make_sequence(640, 480, 6, seed=3)
    sb = {"disparity_now": seq["disparity"][1:], "disparity_prev": seq["disparity"][:-1], "flow": seq["flow"], "t": seq["t"], "q": seq["q"], "dt": seq["dt"]}
    run("stream n=1", cam, synth.Params(dynamic_flow_diff=1, cluster_size=1, neighbor_distance=1), sb, [0, 5])
    run("stream n=10", cam, synth.Params(dynamic_flow_diff=1, cluster_size=40, neighbor_distance=10), sb, [2])
    # 3. quantised inputs: medians tie between different vectors -> the introsort replay, narrowing rounds on degenerate norms
    cam, bq = synth.make_batch(386, 333, 3, seed=77)
    bq["flow"] = (np.round(bq["flow"] * 2) / 2).astype(np.float32)
    bq["disparity_now"] = np.round(bq["disparity_now"]).astype(np.float32)
    run("tied medians", cam, synth.Params(dynamic_flow_diff=1, cluster_size=200, dynamic_speed=0.01), bq, [0, 1, 2])
    # 4. odd sizes: partial tiles at the right and bottom edges, width not a multiple of 4
    cam, bo = synth.make_batch(333, 187, 2, seed=5)
    run("odd size", cam, synth.Params(dynamic_flow_diff=1, cluster_size=5, depth_diff=0.01), bo, [0, 1])
    # 5. one tied cluster whose bounding box needs several runs of the tie replay's cell table, clusterer alone, NO labels plane:
    #    (8256, 8) is the shape and the pass that aborted on 2026-10-04 10:53 (the tie replay's image-scan fallback
    #    still read the labels plane, which this pass does not pass); (2304, 260) has several 64-row segments per column
    from moving_object_detector_amd.pipeline import PLANES as PL
    for W, H in ((8256, 8), (2304, 260)):
        rng = np.random.default_rng(9)
        dyn = rng.random((H, W)) < 0.97
        dyn[:, :3] = False
        base = rng.uniform(0.4, 1.5, size=(2, 3)).astype(np.float32)
        v = base[rng.integers(0, 2, size=(H, W))] * rng.choice(np.array([-1.0, 1.0], np.float32), size=(H, W, 3))
        v[~dyn] = 0.0
        ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
        cloud = {"x": xs * 0.01, "y": ys * 0.01, "z": np.full((H, W), 5.0, np.float32), "vx": np.ascontiguousarray(v[..., 0]),
                 "vy": np.ascontiguousarray(v[..., 1]), "vz": np.ascontiguousarray(v[..., 2])}
        prm = synth.Params(cluster_size=1000, neighbor_distance=4)
        ctx = Context(W, H, max_frames=1, max_objects=W * H // 1000 + 1)
        ctx.set_camera(synth.make_camera(W, H)); ctx.set_params(prm)
        ws = ctx.workspace(1, labels=False)
        for i, k in enumerate(PL):
            ws["planes"][i, 0].copy_(torch.from_numpy(cloud[k]))
        for _ in range(2):
            assert ctx.cluster(1, ws, mask_ready=False) == 0
            ctx.synchronize()
        out = (C.c_uint64 * 96)()
        ctx.lib.mod_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        assert ctx.lib.mod_debug_counters(ctx.h, out) == 0
        counters[:] += np.frombuffer(out, np.uint64)
        lab, ro, K = pyoracle.cluster(cloud, prm, "tidy", max_objects=W * H)
        assert K == 1 and ro[0]["ambiguous"]
        compare_objects(ctx.objects_to_host(ws)[0], ro, strict_velocity=True)
        ctx.close()
        ran.append(f"wide tied cluster {W}x{H}, no labels plane")
    print(json.dumps({"ran": ran, "violations": {str(i - 64): int(v) for i, v in enumerate(counters) if i >= 64}}))


if __name__ == "__main__":
    main()
