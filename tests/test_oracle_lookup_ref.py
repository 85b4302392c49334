"""CPU: the oracle's union-find against the REFERENCE's own LookupTable (scene_flow_clusterer/src/lookup_table.cpp,
the one reference translation unit that compiles here; built into oracle/_ref by oracle/Makefile)."""
import numpy as np
import pytest


def test_fuzz_against_reference_lookup_table(oracle):
    ref = oracle.ref_lookup_lib()
    if ref is None:
        pytest.skip("oracle/_ref/liblookup_ref.so not built (no /root/reference and no prebuilt copy)")
    L = oracle.lib()
    rng = np.random.default_rng(1234)
    for trial in range(40):
        n = int(rng.integers(2, 400))
        a, b = L.orc_lut_create(n), ref.ref_lut_create(n)
        labels = 0
        for _ in range(int(rng.integers(1, 2000))):
            op = rng.random()
            if (op < 0.25 and labels < n) or labels == 0:
                la, lb = L.orc_lut_add_label(a), ref.ref_lut_add_label(b)
                assert la == lb == labels
                labels += 1
            elif op < 0.7:
                x, y = int(rng.integers(0, labels)), int(rng.integers(0, labels))
                L.orc_lut_link(a, x, y)
                ref.ref_lut_link(b, x, y)
            else:
                x = int(rng.integers(0, labels))
                assert L.orc_lut_lookup(a, x) == ref.ref_lut_lookup(b, x)
        for x in range(labels):
            assert L.orc_lut_lookup(a, x) == ref.ref_lut_lookup(b, x)
        # reset() keeps the table storage and restarts the label counter (lookup_table.cpp:12-15)
        L.orc_lut_reset(a)
        ref.ref_lut_reset(b)
        assert L.orc_lut_add_label(a) == ref.ref_lut_add_label(b) == 0
        L.orc_lut_destroy(a)
        ref.ref_lut_destroy(b)


def test_root_is_smallest_label(oracle):
    """link() hooks the larger root under the smaller one, so a set's root is its minimum label — the fact the GPU
    numbering (first_edge_key) relies on."""
    L = oracle.lib()
    rng = np.random.default_rng(7)
    n = 200
    t = L.orc_lut_create(n)
    for _ in range(n):
        L.orc_lut_add_label(t)
    sets = {i: {i} for i in range(n)}
    where = {i: i for i in range(n)}
    for _ in range(300):
        x, y = int(rng.integers(0, n)), int(rng.integers(0, n))
        L.orc_lut_link(t, x, y)
        sx, sy = where[x], where[y]
        if sx != sy:
            sets[sx] |= sets[sy]
            for m in sets[sy]:
                where[m] = sx
            del sets[sy]
    for i in range(n):
        assert L.orc_lut_lookup(t, i) == min(sets[where[i]])
    L.orc_lut_destroy(t)
