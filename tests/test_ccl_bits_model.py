"""CPU: the algorithm of the bit-plane tile kernel (k_ccl_bits) as a plain-Python model on 96-bit integers — closing of gaps, carry-chain
run fill, flood per component, the one-component shortcut, roots / sizes / first_edge_keys / link requests — against brute-force
connected components of the up-left 5 x 5 window graph on random tiles (noise, blobs with holes, bars around the window size, image
borders)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "models"))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_model_matches_brute_force(seed):
    import ccl_bits_model as m
    agg = m.run(seed, 120)
    assert agg["tiles"] > 100 and agg["check"] > 20 and agg["flood"] > 100      # both the shortcut and the flood were exercised


@pytest.mark.parametrize("n", [1, 2, 3, 5, 7, 10])
def test_model_other_window_sizes(n):
    import ccl_bits_model as m
    try:
        agg = m.run(10 + n, 60, n)
        assert agg["tiles"] > 50 and agg["flood"] > 20
    finally:
        m.configure(4)
