"""GPU: tied medians.  Velocities are drawn from sign variants of a few base vectors, so hundreds of members share each
||v|| bit pattern with different vectors: which one std::sort leaves at size/2 is decided by libstdc++'s introsort moves,
which k_median_ties replays.  The object velocity must equal the oracle's (real std::sort) exactly."""
import numpy as np
import pytest

from util import compare_objects

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _tie_cloud(W, H, rng, dyn, n_base):
    base = rng.uniform(0.4, 1.5, size=(n_base, 3)).astype(np.float32)
    pick = rng.integers(0, n_base, size=(H, W))
    sign = rng.choice(np.array([-1.0, 1.0], np.float32), size=(H, W, 3))
    v = base[pick] * sign
    v[~dyn] = 0.0
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    return {"x": xs * 0.01, "y": ys * 0.01, "z": np.full((H, W), 5.0, np.float32), "vx": np.ascontiguousarray(v[..., 0]),
            "vy": np.ascontiguousarray(v[..., 1]), "vz": np.ascontiguousarray(v[..., 2])}


@pytest.mark.parametrize("case", ["one_big_blob", "many_blobs", "tiny_clusters", "single_value"])
def test_tied_median_matches_std_sort(oracle, case):
    from moving_object_detector_amd import synth
    from test_gpu_cluster_stress import _cluster_gpu
    rng = np.random.default_rng({"one_big_blob": 1, "many_blobs": 2, "tiny_clusters": 3, "single_value": 4}[case])
    W, H = 320, 200
    ys, xs = np.mgrid[0:H, 0:W]
    if case == "one_big_blob":
        dyn = (xs > 10) & (xs < 300) & (ys > 5) & (ys < 190) & (rng.random((H, W)) < 0.95)
        prm, nb = synth.Params(cluster_size=1000, neighbor_distance=4), 3
    elif case == "many_blobs":
        dyn = ((xs // 40 + ys // 40) % 2 == 0) & (xs % 40 < 34) & (ys % 40 < 34) & (rng.random((H, W)) < 0.9)
        prm, nb = synth.Params(cluster_size=200, neighbor_distance=2), 5
    elif case == "tiny_clusters":
        dyn = (xs % 8 < 4) & (ys % 8 < 3)                    # 12-pixel clusters: below introsort's partition threshold of 16
        prm, nb = synth.Params(cluster_size=4, neighbor_distance=1), 2
    else:
        dyn = (xs > 20) & (xs < 200) & (ys > 20) & (ys < 150)
        prm, nb = synth.Params(cluster_size=500, neighbor_distance=4), 1   # every member ties
    planes = _tie_cloud(W, H, rng, dyn, nb)
    lab, objs, K = _cluster_gpu(planes, prm, W, H)
    rl, ro, rK = oracle.cluster(planes, prm, "tidy", max_objects=W * H)
    assert K == rK and np.array_equal(lab, rl)
    assert any(o["ambiguous"] for o in ro), "the case is meant to produce ties between different vectors"
    compare_objects(objs, ro, strict_velocity=True)


@pytest.mark.parametrize("W,H", [(8256, 8), (2304, 260)])
def test_tied_median_of_a_very_wide_cluster(oracle, W, H):
    """A flagged cluster whose bounding box has more than 8192 (column x 64-row) cells: k_median_ties lays its members out in
    several runs of 8192 cells (round 1 scanned the labels plane and gave up beyond 2048 columns, keeping the canonical pick)."""
    from moving_object_detector_amd import synth
    from test_gpu_cluster_stress import _cluster_gpu
    rng = np.random.default_rng(W)
    dyn = rng.random((H, W)) < 0.97
    dyn[:, :3] = False
    prm = synth.Params(cluster_size=1000, neighbor_distance=4)
    planes = _tie_cloud(W, H, rng, dyn, 2)
    lab, objs, K = _cluster_gpu(planes, prm, W, H)
    rl, ro, rK = oracle.cluster(planes, prm, "tidy", max_objects=W * H)
    assert K == rK == 1 and np.array_equal(lab, rl)
    assert ro[0]["ambiguous"] and ro[0]["n_points"] > 8192
    compare_objects(objs, ro, strict_velocity=True)
