"""-m gpu: f-4 update_reference on the device (fr3d_update_reference) against the reference's own
arithmetic (BatchMotionCorrector._update_reference, compensate_recording_3D.py:395-429; golden
tests/golden/f4_update_ref.npz written by tools/gen_golden.py) and against the oracle on random cases
with more than 100 volumes (only the last 100 count), float32 inputs and an empty batch."""
import numpy as np
import pytest
from scipy.ndimage import gaussian_filter

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method", ["cubic", "linear"])
def test_update_reference_matches_reference_golden(hip, method):
    from flowreg3d_amd.pipeline import update_reference
    g = golden("f4_update_ref")
    got = update_reference(g["batch_proc"], g["w"], g["ref_proc"], method)
    want = g[f"new_ref_{method}"]
    assert got.dtype == np.float64 and got.shape == want.shape
    # each warped volume is the reference's float32 value up to one ulp (fp64 tap order), the mean is exact fp64
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-7 * float(np.abs(want).max()))
    assert np.abs(got - want).mean() < 1e-9


def test_update_reference_uses_the_last_100_volumes(hip, oracle):
    from flowreg3d_amd.pipeline import update_reference
    rng = np.random.default_rng(8)
    shape, C, T = (5, 9, 8), 1, 103
    bp = np.stack([gaussian_filter(rng.random(shape), 1.0)[..., None] for _ in range(T)], 0)
    w = (rng.standard_normal((T,) + shape + (3,)) * 0.7).astype(np.float32)
    ref = gaussian_filter(rng.random(shape), 1.0)[..., None]
    got = update_reference(bp, w, ref)
    want = oracle.update_reference(bp, w, ref)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-7)
    # dropping the first three volumes changes nothing
    again = update_reference(bp[3:], w[3:], ref)
    assert np.array_equal(got, again)


def test_update_reference_float32_inputs_and_empty_batch(hip, oracle):
    from flowreg3d_amd.pipeline import update_reference
    rng = np.random.default_rng(9)
    shape, C, T = (6, 7, 10), 2, 3
    bp = rng.random((T,) + shape + (C,)).astype(np.float32)
    w = (rng.standard_normal((T,) + shape + (3,)) * 1.5).astype(np.float32)
    ref = rng.random(shape + (C,)).astype(np.float32)
    for method in ("cubic", "linear"):
        got = update_reference(bp, w, ref, method)
        want = oracle.update_reference(bp, w, ref, method)
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-7)
    same = update_reference(bp[:0], w[:0], ref)
    assert np.array_equal(same, ref.astype(np.float64))
    with pytest.raises(ValueError):
        update_reference(bp, w, ref, "nearest")


def test_batch_driver_updates_the_reference_when_asked(hip):
    """options.update_reference (compensate_recording_3D.py:525-526): reference_proc is replaced after every batch."""
    from flowreg3d_amd.pipeline import BatchMotionCorrectorHip, Options
    rng = np.random.default_rng(10)
    shape = (8, 16, 16)
    ref = gaussian_filter(rng.random(shape), 1.5)[..., None]
    video = np.stack([np.roll(ref, t % 2, axis=2) for t in range(4)], 0)
    opt = Options(levels=2, min_level=0, iterations=5, buffer_size=2, weight=[1.0], sigma=[[1.0, 1.0, 1.0, 0.1]],
                  update_reference=True)
    bmc = BatchMotionCorrectorHip(opt)
    bmc.run(video, ref)
    bmc2 = BatchMotionCorrectorHip(Options(levels=2, min_level=0, iterations=5, buffer_size=2, weight=[1.0],
                                           sigma=[[1.0, 1.0, 1.0, 0.1]]))
    bmc2.run(video, ref)
    assert bmc.reference_proc.shape == bmc2.reference_proc.shape
    assert not np.array_equal(bmc.reference_proc, bmc2.reference_proc)
    assert np.isfinite(bmc.reference_proc).all()
