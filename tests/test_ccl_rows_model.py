"""CPU: the algorithm of the one-wave tile kernel (k_ccl_rows), as a plain-Python model, against brute-force connected components
on small multi-tile images: sparse / dense masks, few / many depth levels, NaN depth, neighbor_distance 4, 2, 1."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "models"))


@pytest.mark.parametrize("seed,W,H,n,dens,zlev", [(4, 130, 40, 4, 0.1, 2), (4, 130, 40, 4, 0.3, 8), (23, 70, 33, 2, 0.9, 2), (32, 130, 40, 4, 0.1, 2),
                                                  (9, 70, 33, 4, 0.5, 3), (15, 130, 40, 1, 0.5, 3), (3, 70, 33, 4, 0.9, 2)])
def test_model_matches_brute_force(seed, W, H, n, dens, zlev):
    import ccl_rows_model as m
    assert m.run(seed, W, H, n, dens, zlev) is None
