"""preprocessing.  CPU: the oracle's restatement of normalize +
scipy gaussian_filter against the reference's own outputs (tests/golden/f1_preproc.npz).
GPU (-m gpu): the HIP path (fr3d_preprocess) against the same goldens and the oracle."""
import numpy as np
import pytest

from conftest import golden

TOL = 2e-15  # libm vs NumPy exp() in the kernel weights: 1 ulp


@pytest.mark.parametrize("cn", ["together", "separate"])
def test_oracle_preprocessing_vs_reference(oracle, cn):
    g = golden("f1_preproc")
    n5 = oracle.normalize(g["batch"], ref=g["ref"], channel_normalization=cn)
    assert np.array_equal(n5, g[f"norm5_{cn}"])
    f5 = oracle.apply_gaussian_filter(n5, g["sigma"])
    assert f5.dtype == np.float64 and np.abs(f5 - g[f"filt5_{cn}"]).max() < TOL
    n4 = oracle.normalize(g["ref"], channel_normalization=cn)
    assert np.abs(oracle.apply_gaussian_filter(n4, g["sigma"]) - g[f"filt4_{cn}"]).max() < TOL


def test_oracle_gaussian_matches_scipy_on_short_axes(oracle):
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(0)
    for shape, sig in (((3, 5, 40), (2.0, 0.5, 1.5)), ((6, 7, 8), (0.0, 1.0, 0.1)), ((2, 2, 2), (3.0, 3.0, 3.0))):
        a = rng.random(shape)
        assert np.abs(oracle.gaussian_filter3(a, sig) - gaussian_filter(a, sigma=sig, mode="reflect")).max() < TOL


def test_host_normalize_mirror_matches_reference():
    from flowreg3d_amd import preprocess
    g = golden("f1_preproc")
    for cn in ("together", "separate"):
        assert np.array_equal(preprocess.normalize(g["batch"], ref=g["ref"], channel_normalization=cn),
                              g[f"norm5_{cn}"])


@pytest.mark.gpu
@pytest.mark.parametrize("cn", ["together", "separate"])
def test_gpu_preprocess_frames_vs_reference(hip, cn):
    from flowreg3d_amd import preprocess
    g = golden("f1_preproc")
    out = preprocess.preprocess_frames(g["batch"], normalization_ref=g["ref"], sigma=g["sigma"],
                                       channel_normalization=cn)
    assert out.dtype == np.float64 and out.shape == g["batch"].shape
    assert np.abs(out - g[f"filt5_{cn}"]).max() < TOL
    out4 = preprocess.preprocess_frames(g["ref"], sigma=g["sigma"], channel_normalization=cn)
    assert np.abs(out4 - g[f"filt4_{cn}"]).max() < TOL
    out32 = preprocess.preprocess_frames(g["batch"], normalization_ref=g["ref"], sigma=g["sigma"],
                                         channel_normalization=cn, out_float32=True)
    assert out32.dtype == np.float32 and np.array_equal(out32, g[f"filt5_{cn}"].astype(np.float32))


@pytest.mark.gpu
def test_gpu_preprocess_uint16_and_filter_only(hip, oracle):
    from flowreg3d_amd import preprocess
    g = golden("f1_preproc")
    out = preprocess.preprocess_frames(g["batch_u16"], normalization_ref=g["ref"], sigma=np.array([1.0, 1.0, 1.0, 0.1]))
    assert np.abs(out - g["filt5_u16"]).max() < TOL
    rng = np.random.default_rng(3)
    a = rng.random((5, 6, 70, 1))
    got = preprocess.apply_gaussian_filter(a, np.array([2.5, 0.0, 1.0]))  # sy = 0: axis skipped
    want = oracle.apply_gaussian_filter(a, np.array([2.5, 0.0, 1.0]))
    assert np.abs(got - want).max() < TOL
    with pytest.raises(RuntimeError):  # scipy raises RuntimeError("boundary mode not supported") too
        preprocess.apply_gaussian_filter(a, np.ones(3), mode="no-such-mode")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["reflect", "constant", "nearest", "mirror", "wrap", "grid-mirror", "grid-constant",
                                  "grid-wrap"])
def test_gpu_gaussian_filter_boundary_modes_match_scipy(hip, mode):
    """apply_gaussian_filter hands `mode` to scipy.ndimage.gaussian_filter (util/image_processing_3D.py:95-162): every
    boundary mode of scipy, against scipy itself -- volumes with axes shorter than the kernel radius (several
    reflections / wraps), a 5-D batch with a temporal sigma, per-channel sigmas"""
    from scipy.ndimage import gaussian_filter
    from flowreg3d_amd import preprocess
    rng = np.random.default_rng(7)
    for shape, sigma in (((9, 12, 31, 2), np.array([1.0, 1.5, 0.8])),
                         ((3, 2, 40, 1), np.array([2.0, 2.0, 2.0])),          # axes of 2 and 3 under radius 8
                         ((1, 17, 6, 1), np.array([1.0, 1.0, 1.0])),          # an axis of length 1
                         ((4, 6, 7, 9, 2), np.array([[1.0, 0.7, 1.2, 0.9], [0.5, 1.0, 1.0, 0.0]]))):
        a = rng.random(shape)
        got = preprocess.apply_gaussian_filter(a, sigma, mode=mode)
        want = np.empty_like(a)
        for c in range(shape[-1]):
            s = sigma[min(c, len(sigma) - 1)] if sigma.ndim == 2 else sigma
            s = tuple(s[::-1]) if a.ndim == 5 else (s[2], s[1], s[0])
            want[..., c] = gaussian_filter(a[..., c], sigma=s, mode=mode, truncate=4.0)
        assert got.dtype == np.float64
        assert np.abs(got - want).max() < 4e-15, (mode, shape, np.abs(got - want).max())
