"""-m gpu: the VERIFICATION mode (fr3d_get_displacement_verify, k_verify.hip) against the CPU oracle built with the same
portable pow (oracle `ppow` build; tests/test_portable_pow.py pins that build to the default one and the pow to libm).

The verification mode runs the reference's arithmetic, operation by operation, on the engine's data path: compact
skewed solver layout and its tables, hyperplane launch schedule and tile decode, resampler, prefilter, gather, tensor,
fp64 median.  The bar is BIT-IDENTITY of the whole get_displacement result -- float64 arrays compared with
np.array_equal -- on small cases here (the oracle runs beside the GPU) and on full-size samples committed as
tests/golden/fullsize_*_ppow.npz (tools/gen_fullsize_golden.py --ppow).  What that establishes: every stage, the
skewed indexing and the launch schedule are exact at any size; what the shipped solver modes differ from the CPU path
by (1e-5 ... 3e-4 voxels, tests/test_gpu_fullsize_parity.py) is rounding of their reformulated update, nothing else."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

KW = dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=30, min_level=0, levels=3, eta=0.8, a_smooth=1.0, a_data=0.45)


@pytest.fixture()
def ppow_oracle(oracle):
    oracle.use_build("ppow")
    yield oracle
    oracle.use_build("")


def _same(a, b, label):
    d = np.abs(a - b)
    assert np.array_equal(a, b), f"{label}: not bit-identical: {int((d > 0).sum())} of {d.size} values differ, max |diff| {d.max():.3e}"


@pytest.mark.parametrize("shape,ch,kw", [
    ((20, 28, 30), 1, {}),
    ((33, 41, 66), 1, dict(iterations=47, update_lag=4)),        # rows of every length, partly filled tiles
    ((24, 40, 36), 2, dict(a_data=[0.45, 0.6])),                 # two channels: per-channel accumulation order
    ((16, 30, 30), 3, dict(update_lag=1, iterations=12)),        # psi update on every iteration
    ((40, 24, 26), 1, dict(min_level=1, levels=4)),              # final resample back to full size
    ((18, 22, 70), 1, dict(a_data=1.0, eta=0.7)),                # a_data = 1: psi is not used
    # the psi_smooth branch (level_solver_3d.py:262-311, 400-471; get_displacement's own default a_smooth = 0.5): psi_smooth on
    # the padded grid from the increments of t-1 (interior) and t-2 (ghost ring), psi-weighted stencil in the reference's order
    ((20, 28, 30), 1, dict(a_smooth=0.5)),
    ((19, 33, 26), 2, dict(a_smooth=0.5, a_data=[0.45, 0.6], update_lag=3, iterations=17)),
    ((7, 9, 40), 1, dict(a_smooth=0.8, levels=2)),                # thin volume: every voxel touches a face
    ((24, 26, 22), 1, dict(a_smooth=0.5, min_level=1, levels=4, eta=0.75)),
])
def test_verify_mode_is_bit_identical_to_the_ppow_oracle(hip, ppow_oracle, shape, ch, kw):
    from flowreg3d_amd.synthetic import make_pair
    fixed, moving, _ = make_pair(shape, seed=11, channels=ch, scale=0.8)
    args = dict(KW, **kw)
    want = ppow_oracle.get_displacement(fixed, moving, **args)
    got = hip.get_displacement_verify(fixed, moving, **args)
    assert got.dtype == np.float64 and got.shape == want.shape
    _same(got, want, f"{shape} C={ch} {kw}")


def test_verify_mode_with_initial_flow_and_voxel_weights(hip, ppow_oracle):
    from flowreg3d_amd.synthetic import make_pair
    shape = (22, 30, 34)
    fixed, moving, gt = make_pair(shape, seed=4, channels=2)
    rng = np.random.default_rng(2)
    uvw = (0.6 * gt + 0.05 * rng.standard_normal(gt.shape)).astype(np.float32)
    weight = rng.uniform(0.2, 1.0, shape + (2,)).astype(np.float32)
    want = ppow_oracle.get_displacement(fixed, moving, uvw=uvw.copy(), weight=weight, **KW)
    got = hip.get_displacement_verify(fixed, moving, uvw=uvw.copy(), weight=weight, **KW)
    _same(got, want, "uvw + 4-D weight")


def test_shipped_modes_differ_from_the_verified_path_by_rounding_only(hip):
    """the fp64-storage mode against the verification mode on one case: ~1e-7 (exact level tail; what is left is the fp32 rounding of the output); fp32 storage 4e-5"""
    from flowreg3d_amd.synthetic import make_pair
    fixed, moving, _ = make_pair((24, 40, 36), seed=11, channels=1)
    v = hip.get_displacement_verify(fixed, moving, **KW)
    for mode, tol in ((2, 1e-6), (1, 8e-5)):
        f = hip.get_displacement(fixed, moving, solver_fp64=mode, **KW)
        d = np.linalg.norm(f - v, axis=-1)
        print(f"mode {mode} vs verification mode: mean {d.mean():.2e} max {d.max():.2e}")
        assert d.mean() < tol, (mode, d.mean())


@pytest.mark.parametrize("case", ["cfg5", "cfg3", "cfg2_asmooth05"])
def test_fullsize_verify_mode_is_bit_identical_to_the_ppow_oracle_sample(hip, case):
    """BASELINE configurations 5 and 3 AT FULL SIZE, and config 2's volume with a_smooth = 0.5 (the psi_smooth branch): the
    verification mode against the `ppow` oracle's committed sample (float64 lattice of every 8th voxel + central 32^3
    block), bit for bit."""
    from flowreg3d_amd.synthetic import fullsize_case
    path = os.path.join(GOLDEN, f"fullsize_{case}_ppow.npz")
    if not os.path.exists(path):
        pytest.skip(f"no committed ppow-oracle sample for {case}")
    g = np.load(path)
    meta = json.loads(bytes(g["meta"]).decode())
    fixed, moving, gt, kw = fullsize_case(case)
    flow = hip.get_displacement_verify(fixed, moving, **{k: v for k, v in kw.items()})
    st, bl = meta["stride"], meta["block"]
    z0, y0, x0 = meta["block_origin_zyx"]
    assert g["lattice"].dtype == np.float64
    _same(flow[::st, ::st, ::st], g["lattice"], f"{case} lattice")
    _same(flow[z0:z0 + bl, y0:y0 + bl, x0:x0 + bl], g["block"], f"{case} block")


def test_portable_pow_gives_the_same_bits_on_device_and_host(hip):
    """the premise of the verification mode: portable_pow.h compiled by hipcc for gfx950 and by gcc for the host agree
    bit for bit -- 20 million arguments over the ranges the psi nonlinearities feed it"""
    import ctypes as C
    import sys
    from flowreg3d_amd import _lib
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_portable_pow import _build
    host = _build()
    lib = _lib.init(0)
    rng = np.random.default_rng(5)
    n = 5_000_000
    dp = C.POINTER(C.c_double)
    for k, (x, y) in enumerate([(10.0 ** rng.uniform(-6, 8, n), rng.uniform(-1.0, 0.0, n)),
                                (1e-6 + rng.random(n) * 1e-4, np.full(n, -0.55)),
                                (1e-6 + 10.0 ** rng.uniform(-9, 3, n), np.full(n, -0.55)),
                                (1.0 + rng.normal(0, 0.3, n) ** 2, rng.choice([-0.55, -0.5, -0.9, -0.1], n))]):
        x = np.ascontiguousarray(x); y = np.ascontiguousarray(y)
        a = np.empty(n); b = np.empty(n)
        host.ppow_many(x.ctypes.data_as(dp), y.ctypes.data_as(dp), n, a.ctypes.data_as(dp))
        _lib.check(lib.fr3d_portable_pow(_lib.ptr(x), _lib.ptr(y), n, _lib.ptr(b)))
        bad = np.flatnonzero(a != b)
        assert bad.size == 0, (k, bad.size, x[bad[:3]], y[bad[:3]], a[bad[:3]], b[bad[:3]])
