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
