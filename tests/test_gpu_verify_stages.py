"""-m gpu: the stages of the verification mode one by one, float64 against the CPU oracle, BIT FOR BIT
(np.array_equal) -- what makes the whole-pipeline bit-identity of tests/test_gpu_verify_mode.py debuggable:

  * cubic B-spline coefficients of the pad-free prefilter (fr3d_spline_coefficients) against the oracle's filter of
    the 12-voxel-padded array.  This is the test that would have caught the engine's spline pole: the correctly
    rounded value of sqrt(3) - 2 instead of SciPy's `sqrt(3.0) - 2.0` in double arithmetic (two ulp apart; 94 % of
    the coefficients differed in their last bits, one warped voxel in four million changed its float32 value);
  * the float64 gradient-constancy motion tensor (fr3d_motion_tensor_f64), unit and non-unit spacings;
  * the reference-order sweep alone (fr3d_level_solve_verify) against the ppow oracle's compute_flow_3d.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(48, 64, 80), (41, 90, 130)])
def test_spline_coefficients_are_bit_identical_to_the_oracle(hip, oracle, shape):
    from flowreg3d_amd import _lib
    from flowreg3d_amd.synthetic import make_pair
    lib = _lib.init(0)
    _, moving, _ = make_pair(shape, seed=7, cheap=True)
    Z, Y, X = shape
    want = oracle.spline_filter3(np.pad(moving.astype(np.float64), 12, mode="edge"))[10:-10, 10:-10, 10:-10]
    got = np.empty((Z + 4, Y + 4, X + 4), np.float64)
    vol = np.ascontiguousarray(moving, np.float32)
    _lib.check(lib.fr3d_spline_coefficients(_lib.ptr(vol), Z, Y, X, _lib.ptr(got)))
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {got.size} coefficients differ"


@pytest.mark.parametrize("h", [(1.0, 1.0, 1.0), (1.25, 1.3, 1.1)])
def test_fp64_motion_tensor_is_bit_identical_to_the_oracle(hip, oracle, h):
    from flowreg3d_amd import _lib
    from flowreg3d_amd.synthetic import make_pair
    lib = _lib.init(0)
    shape = (30, 44, 52)
    fixed, moving, _ = make_pair(shape, seed=3, cheap=True)
    Z, Y, X = shape
    J = oracle.get_motion_tensor_gc(fixed, moving, *h)
    want = np.stack([j[1:-1, 1:-1, 1:-1] for j in J])
    got = np.empty((10, Z, Y, X), np.float64)
    f1, f2 = np.ascontiguousarray(fixed, np.float32), np.ascontiguousarray(moving, np.float32)
    _lib.check(lib.fr3d_motion_tensor_f64(_lib.ptr(f1), _lib.ptr(f2), Z, Y, X, *h, _lib.ptr(got)))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("channels,its,lag", [(1, 17, 5), (2, 9, 2), (3, 6, 1)])
def test_reference_order_sweep_is_bit_identical_to_the_ppow_oracle(hip, oracle, channels, its, lag):
    from flowreg3d_amd import _lib
    from flowreg3d_amd.synthetic import make_pair
    lib = _lib.init(0)
    shape = (26, 70, 45)
    fixed, moving, gt = make_pair(shape, seed=9, channels=channels, cheap=True)
    if channels == 1:
        fixed, moving = fixed[..., None], moving[..., None]
    Z, Y, X = shape
    Js = [oracle.get_motion_tensor_gc(fixed[..., c], moving[..., c], 1.0, 1.0, 1.0) for c in range(channels)]
    rng = np.random.default_rng(1)
    uvw = (0.8 * np.moveaxis(gt, -1, 0) + 0.02 * rng.standard_normal((3, Z, Y, X))).astype(np.float32)
    wch = rng.uniform(0.3, 1.0, (channels, Z, Y, X)).astype(np.float32)
    pad = lambda a: np.pad(a.astype(np.float64), 1, mode="edge")
    wt = np.zeros((Z + 2, Y + 2, X + 2, channels))
    wt[1:-1, 1:-1, 1:-1] = np.moveaxis(wch.astype(np.float64), 0, -1)
    a_data = [0.45, 0.6, 0.3][:channels]
    try:
        oracle.use_build("ppow")
        want = oracle.compute_flow_3d(*[np.stack([Js[c][q] for c in range(channels)], -1) for q in range(10)], wt,
                                      pad(uvw[0]), pad(uvw[1]), pad(uvw[2]), 0.3, 0.25, 0.2, its, lag, a_data, 1.0,
                                      1.0, 1.0, 1.0)[1:-1, 1:-1, 1:-1]
    finally:
        oracle.use_build("")
    Jin = np.ascontiguousarray(np.stack([np.stack([Js[c][q][1:-1, 1:-1, 1:-1] for q in range(10)]) for c in range(channels)]))
    out = np.empty((3, Z, Y, X), np.float64)
    al = (C.c_double * 3)(0.3, 0.25, 0.2)
    ad = (C.c_double * channels)(*a_data)
    _lib.check(lib.fr3d_level_solve_verify(_lib.ptr(Jin), _lib.ptr(wch), _lib.ptr(uvw), Z, Y, X, channels, al, its, lag, ad,
                                           1.0, 1.0, 1.0, _lib.ptr(out)))
    assert np.array_equal(np.moveaxis(out, 0, -1), want)
