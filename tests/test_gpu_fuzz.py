"""-m gpu: randomised shapes (axes of length 1..70), channel counts, pyramid and solver parameters,
initial flows and weights -- the GPU path (fp64 solver storage) against the CPU oracle, including
inputs that both must reject.  tools/fuzz_vs_oracle.py runs the same generator with more cases."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes_and_parameters_match_the_oracle(hip, oracle):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_vs_oracle
    bad, worst = fuzz_vs_oracle.run(n_cases=40, seed=11, verbose=False)
    assert bad == 0 and worst < 1e-4, (bad, worst)


def test_random_shapes_and_parameters_in_the_default_solver_mode(hip, oracle):
    """The same generator (1..6 channels, axes of length 1..70) with the solver mode left to the library
    (FR3D_SOLVER_AUTO: fp32 solver storage + fp64 update arithmetic for one channel -- the benched mode --
    and fp64 storage for several); bound 2e-4 * max(1, |flow|max) on the mean end-point error."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_vs_oracle
    bad, worst = fuzz_vs_oracle.run(n_cases=40, seed=23, verbose=False, mode=None)
    assert bad == 0 and worst < 2e-4, (bad, worst)


def test_random_shapes_and_parameters_in_packed_storage(hip, oracle):
    """packed 42-bit solver storage forced on the same generator (tiny volumes, 1..6 channels, a_smooth 1 and 0.5 --
    the latter falls back to fp64 storage)"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_vs_oracle
    bad, worst = fuzz_vs_oracle.run(n_cases=30, seed=31, verbose=False, mode=3)
    assert bad == 0 and worst < 2e-4, (bad, worst)


@pytest.mark.parametrize("mode", [2, 3])
def test_random_shapes_and_parameters_through_the_window_sweep(hip, oracle, mode):
    """the same generator with fr3d_params.solver_sweep = FR3D_SWEEP_WINDOW: one- and two-channel cases with a_smooth = 1
    run the window kernel (k_sor_win.hip), everything else falls back to the plane sweep -- against the CPU oracle"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_vs_oracle
    bad, worst = fuzz_vs_oracle.run(n_cases=30, seed=59 + mode, verbose=False, mode=mode, sweep=2)
    assert bad == 0 and worst < (1e-4 if mode == 2 else 2e-4), (bad, worst)


def test_verification_mode_with_psi_smooth_is_bit_identical_on_random_cases(hip, oracle):
    """the verification mode on the random generator WITH a_smooth drawn from {1, 1, 0.5} (round 4: the psi_smooth
    branch is covered): np.array_equal against the ppow oracle on every case"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_vs_oracle
    bad, worst = fuzz_vs_oracle.run(n_cases=24, seed=71, verbose=False, mode="verify", verify_smooth=True)
    assert bad == 0, (bad, worst)


def test_verification_mode_is_bit_identical_on_random_shapes_and_parameters(hip, oracle):
    """the verification mode against the oracle's ppow build on the random generator (axes of length 1..70, 1..6
    channels, random pyramid / solver parameters, initial flows, weights): np.array_equal on every case, and the
    inputs the reference rejects are rejected"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_vs_oracle
    bad, worst = fuzz_vs_oracle.run(n_cases=40, seed=47, verbose=False, mode="verify")
    assert bad == 0, (bad, worst)


@pytest.mark.parametrize("C", [4, 5, 8])
def test_many_channels_match_the_oracle(hip, oracle, C):
    """C = 4 (last unrolled instantiation), 5 and 8 = FR3D_MAX_CHANNELS (channel loop bound at run time):
    level_solver_3d.py:356-377 loops over any channel count."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(40 + C)
    shape = (10, 18, 14)
    def vol():
        a = gaussian_filter(rng.random(shape), 1.0, mode="reflect")
        return ((a - a.min()) / (a.max() - a.min())).astype(np.float32)
    fixed = np.stack([vol() for _ in range(C)], -1)
    moving = np.stack([0.97 * gaussian_filter(fixed[..., c], 0.6) + 0.02 for c in range(C)], -1).astype(np.float32)
    w = rng.uniform(0.2, 1.0, C)
    for a_smooth in (1.0, 0.5):
        kw = dict(alpha=(0.3, 0.25, 0.2), update_lag=3, iterations=8, min_level=0, levels=3, eta=0.8, a_smooth=a_smooth,
                  a_data=0.45, weight=w)
        want = oracle.get_displacement(fixed, moving, **kw)
        got = hip.get_displacement(fixed, moving, **kw)
        epe = np.linalg.norm(got - want, axis=-1)
        assert epe.mean() < 1e-4, (C, a_smooth, epe.mean(), epe.max())


def test_eta_one_runs_like_the_reference(hip, oracle):
    """eta == 1.0 is legal in the reference (OFOptions allows eta <= 1): every pyramid level has the full size."""
    rng = np.random.default_rng(2)
    from scipy.ndimage import gaussian_filter
    fixed = gaussian_filter(rng.random((8, 12, 12)), 1.0).astype(np.float32)
    moving = np.roll(fixed, 1, axis=2)
    kw = dict(alpha=(0.5, 0.5, 0.5), update_lag=3, iterations=5, min_level=0, levels=3, eta=1.0, a_smooth=1.0, a_data=0.45)
    want = oracle.get_displacement(fixed, moving, **kw)
    got = hip.get_displacement(fixed, moving, solver_fp64=2, **kw)
    assert np.linalg.norm(got - want, axis=-1).mean() < 1e-4


def test_level_rounded_to_zero_is_rejected_like_the_reference(hip, oracle):
    """An axis of length 1 with eta = 0.5 rounds to 0 on the first coarse level; the reference raises
    ZeroDivisionError in its resampler (util/resize_util_3D.py:116-128)."""
    rng = np.random.default_rng(0)
    fixed = rng.random((1, 6, 70)).astype(np.float32)
    kw = dict(alpha=(1.0, 1.0, 1.0), update_lag=5, iterations=4, min_level=3, levels=9, eta=0.5, a_smooth=1.0,
              a_data=0.45)
    with pytest.raises(ValueError):
        oracle.get_displacement(fixed, fixed, **kw)
    with pytest.raises((ValueError, RuntimeError)):
        hip.get_displacement(fixed, fixed, **kw)


def test_random_stage_inputs_match_the_oracle(hip, oracle):
    """Resampler (arbitrary source/target sizes, bit-exact), cubic and linear warp (displacements from
    sub-voxel to far outside the volume), 5^3 median on shapes down to a single voxel."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_stages
    assert fuzz_stages.run(n_cases=40, seed=5, verbose=False) == 0


def test_random_preprocessing_inputs_match_the_oracle(hip, oracle):
    """normalize + Gaussian filter on the device for random dtypes (u8/u16/i16/f32/f64), shapes with
    and without a time axis, per-channel sigmas and both normalisation modes."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_preprocess
    assert fuzz_preprocess.run(n_cases=40, seed=3, verbose=False) == 0
