"""CPU: the synthetic generator is deterministic and contains the value classes the reference branches on."""
import numpy as np


def test_deterministic_and_distinct():
    from moving_object_detector_amd import synth
    _, a = synth.make_frame(160, 120, seed=5, frame=3)
    _, b = synth.make_frame(160, 120, seed=5, frame=3)
    _, c = synth.make_frame(160, 120, seed=5, frame=4)
    assert np.array_equal(a.disparity_now, b.disparity_now, equal_nan=True) and np.array_equal(a.flow, b.flow, equal_nan=True)
    assert not np.array_equal(a.disparity_now, c.disparity_now, equal_nan=True)


def test_value_classes_present():
    from moving_object_detector_amd import synth
    cam, f = synth.make_frame(640, 480, seed=1, frame=0)
    for d in (f.disparity_now, f.disparity_prev):
        assert np.isnan(d).any() and (d == 0).any() and (d < 0).any() and (d > cam.max_disparity).any()
    assert np.isnan(f.flow).any() and (np.abs(np.nan_to_num(f.flow)) > 1e5).any()
    assert abs(np.linalg.norm(f.quaternion) - 1.0) < 1e-12


def test_batch_layout():
    from moving_object_detector_amd import synth
    cam, b = synth.make_batch(96, 64, 3, seed=2)
    assert b["disparity_now"].shape == (3, 64, 96) and b["flow"].shape == (3, 64, 96, 2) and b["q"].shape == (3, 4)
    assert b["disparity_now"].dtype == np.float32 and b["t"].dtype == np.float64


def test_staggered_planes_do_not_overlap_and_keep_their_stagger():
    """pipeline.staggered(): planes carved from one block, consecutive bases 1 MiB further apart than the plane before is long
    (back-to-back planes of 512 x 1280 x 720 floats are multiples of 8 MiB apart)."""
    import torch
    from moving_object_detector_amd.pipeline import PLANE_STAGGER_BYTES, staggered
    sizes = [1000, 1000, 2000]
    a, b, c = staggered(sizes, torch.float32, "cpu")
    assert [t.numel() for t in (a, b, c)] == sizes
    assert b.data_ptr() - a.data_ptr() == 4 * sizes[0] + PLANE_STAGGER_BYTES
    assert c.data_ptr() - b.data_ptr() == 4 * sizes[1] + PLANE_STAGGER_BYTES
    a.fill_(1.0); b.fill_(2.0); c.fill_(3.0)
    assert float(a.sum()) == 1000.0 and float(b.sum()) == 2000.0 and float(c.sum()) == 6000.0
