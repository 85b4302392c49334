"""CPU: the two restatements of the SGM disparity estimator (oracle/sgm_ref.cpp, oracle/sgm_numpy.py) against each other, stage
by stage, and against the committed fixture.  Integer arithmetic: everything is compared exactly."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "sgm_320x240.npz")


@pytest.mark.parametrize("W,H,D,seed", [(96, 64, 32, 1), (131, 77, 128, 3), (40, 9, 64, 4), (9, 7, 16, 5), (8, 6, 8, 6)])
def test_restatements_agree_stage_by_stage(W, H, D, seed):
    from oracle import pysgm
    from oracle import sgm_numpy as sn
    left, right, _ = sn.make_stereo(W, H, seed, D, n_boxes=2)
    st = sn.compute(left, right, D, stages=True)
    assert np.array_equal(pysgm.census(left), st["census_left"]) and np.array_equal(pysgm.census(right), st["census_right"])
    C = pysgm.cost(st["census_left"], st["census_right"], D)
    assert np.array_equal(C, st["cost"])
    for i in range(8):
        assert np.array_equal(pysgm.aggregate(C, 6, 96, i), st["paths"][i]), i
    disp, S = pysgm.compute(left, right, D, want_S=True)
    assert np.array_equal(S, st["S"]) and np.array_equal(disp, st["disparity"])
    for kw in (dict(paths=4), dict(lr_check=False), dict(median=False), dict(P1=3, P2=40)):
        assert np.array_equal(pysgm.compute(left, right, D, **kw), sn.compute(left, right, D, **kw)), kw


def test_census_known_answers():
    """Hand-checkable words: a horizontal ramp makes every pair (p, -p) with dx < 0 'less', dx > 0 'greater or equal'."""
    from oracle import pysgm
    img = np.tile(np.arange(32, dtype=np.uint8), (12, 1))
    c = pysgm.census(img)
    assert (c[:3] == 0).all() and (c[-3:] == 0).all() and (c[:, :4] == 0).all() and (c[:, -4:] == 0).all()
    # rows dy = -3..-1: dx = -4..4 -> bits 0000 1 1111 (dx = 0 compares equal: 1); row dy = 0: dx = -4..-1 -> 0000
    row = 0b000011111
    want = (row << 22) | (row << 13) | (row << 4)
    assert (c[3:-3, 4:-4] == want).all()
    assert (pysgm.census(np.full((12, 32), 7, np.uint8))[3:-3, 4:-4] == 0x7FFFFFFF).all()      # flat image: every pair equal


def test_fixture():
    from oracle import pysgm
    g = np.load(GOLD)
    D, P1, P2 = int(g["D"]), int(g["P1"]), int(g["P2"])
    cl, cr = pysgm.census(g["left"]), pysgm.census(g["right"])
    assert np.array_equal(cl, g["census_left"]) and np.array_equal(cr, g["census_right"])
    C = pysgm.cost(cl, cr, D)
    assert np.array_equal(C[g["rows"]], g["cost_rows"])
    for i, k in ((i, f"path{i}") for i in range(8)):        # all eight aggregation paths are pinned
        L = pysgm.aggregate(C, P1, P2, i)
        assert np.array_equal(L[g["rows"]], g[k + "_rows"]) and np.array_equal(L.astype(np.uint32).sum(axis=2), g[k + "_sum"])
    disp, S = pysgm.compute(g["left"], g["right"], D, P1, P2, want_S=True)
    assert np.array_equal(S.astype(np.uint32).sum(axis=2), g["S_sum"]) and np.array_equal(disp, g["disparity"])
    v = disp >= 0
    assert v.mean() > 0.7 and (np.abs(disp - g["truth"])[v] <= 1).mean() > 0.8       # it does estimate the scene's disparity
