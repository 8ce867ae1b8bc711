"""-m gpu: edge cases of the hot path against the oracle -- odd and degenerate shapes, levels whose
smallest side is <= 5 (median skipped, core/optical_flow_3d.py:517), min_level clamping, initial
flow, spatially varying weights, repeated calls with changing sizes, argument errors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KW = dict(alpha=(0.25, 0.3, 0.35), update_lag=3, iterations=12, min_level=0, levels=3, eta=0.8, a_smooth=1.0,
          a_data=0.45)


def _pair(shape, seed=0, C=1):
    rng = np.random.default_rng(seed)
    from scipy.ndimage import gaussian_filter
    def vol():
        a = gaussian_filter(rng.random(shape), 1.2, mode="reflect")
        return ((a - a.min()) / (a.max() - a.min() + 1e-12)).astype(np.float32)
    f = np.stack([vol() for _ in range(C)], -1)
    m = np.stack([np.roll(f[..., c], 1, axis=min(2, f[..., c].ndim - 1)) * 0.98 + 0.01 for c in range(C)], -1)
    if C == 1:
        f, m = f[..., 0], m[..., 0]
    return f, m.astype(np.float32)


def _epe(a, b):
    d = np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64), axis=-1)
    return d.mean(), d.max()


@pytest.mark.parametrize("shape", [(5, 40, 40), (9, 9, 9), (3, 30, 50), (1, 24, 24), (17, 1, 33), (12, 65, 7),
                                   (33, 20, 129)])
def test_odd_shapes_match_oracle(hip, oracle, shape):
    fixed, moving = _pair(shape, seed=sum(shape))
    want = oracle.get_displacement(fixed, moving, **KW)
    got = hip.get_displacement(fixed, moving, solver_fp64=2, **KW)
    got32 = hip.get_displacement(fixed, moving, **KW)
    assert got.shape == shape + (3,)
    assert np.isfinite(got).all() and np.isfinite(got32).all()
    m64, _ = _epe(got, want)
    m32, _ = _epe(got32, want)
    assert m64 < 1e-4, (shape, m64)
    assert m32 < 1e-3, (shape, m32)


@pytest.mark.parametrize("min_level,levels", [(1, 50), (2, 50), (7, 3), (0, 1)])
def test_min_level_and_level_clamping(hip, oracle, min_level, levels):
    fixed, moving = _pair((16, 24, 28), seed=3)
    kw = dict(KW, min_level=min_level, levels=levels)
    want = oracle.get_displacement(fixed, moving, **kw)
    got = hip.get_displacement(fixed, moving, solver_fp64=2, **kw)
    assert got.shape == want.shape
    assert _epe(got, want)[0] < 1e-4


def test_initial_flow_and_spatial_weight(hip, oracle):
    fixed, moving = _pair((14, 20, 22), seed=5, C=2)
    rng = np.random.default_rng(1)
    from scipy.ndimage import gaussian_filter
    uvw = np.stack([gaussian_filter(rng.standard_normal((14, 20, 22)), 2.0) for _ in range(3)], -1)
    w3 = 0.2 + rng.random((14, 20, 22))          # 3-D weight: broadcast to both channels (:376-381)
    w4 = np.stack([w3, 1.5 - w3], -1)             # 4-D weight: used as is
    for weight in (w3, w4, np.array([2.0]), np.array([1.0, 3.0, 5.0])):
        want = oracle.get_displacement(fixed, moving, uvw=uvw.copy(), weight=weight, **KW)
        got = hip.get_displacement(fixed, moving, uvw=uvw.copy(), weight=weight, solver_fp64=2, **KW)
        assert _epe(got, want)[0] < 1e-4


def test_zero_iterations_and_identical_volumes(hip, oracle):
    fixed, moving = _pair((10, 16, 16), seed=9)
    kw = dict(KW, iterations=0)
    assert np.array_equal(hip.get_displacement(fixed, moving, **kw), np.zeros((10, 16, 16, 3)))
    flow = hip.get_displacement(fixed, fixed, **KW)
    assert np.abs(flow).max() < 1e-5  # nothing to register


def test_alternating_sizes_reuse_the_workspace(hip, oracle):
    a = _pair((12, 18, 20), seed=1)
    b = _pair((20, 30, 26), seed=2)
    ra = oracle.get_displacement(*a, **KW)
    rb = oracle.get_displacement(*b, **KW)
    for _ in range(2):
        assert _epe(hip.get_displacement(*a, solver_fp64=2, **KW), ra)[0] < 1e-4
        assert _epe(hip.get_displacement(*b, solver_fp64=2, **KW), rb)[0] < 1e-4


def test_scalar_displacements_in_imregister(hip, oracle):
    # xcorr pre-alignment calls imregister_wrapper with scalar shifts (tests/util/test_xcorr_prealignment.py:50)
    rng = np.random.default_rng(4)
    vol = rng.random((8, 12, 14)).astype(np.float32)
    ref = rng.random((8, 12, 14)).astype(np.float32)
    for method in ("linear", "cubic"):
        got = hip.imregister_wrapper(vol, 1.25, -0.5, 0.75, ref, method)
        want = oracle.imregister_wrapper(vol, 1.25, -0.5, 0.75, ref, method)
        assert np.abs(got - want).max() <= 1.2e-7


def test_c_abi_rejects_bad_arguments(hip):
    import ctypes as C
    from flowreg3d_amd import _lib
    lib = _lib.init(0)
    p = _lib.make_params((0.25,) * 3, 5, 10, 0, 3, 0.8, 1.0, 0.45, 1)
    z = np.zeros((4, 4, 4), np.float32)
    out = np.zeros((4, 4, 4, 3), np.float32)
    assert lib.fr3d_get_displacement(C.byref(p), _lib.ptr(z), _lib.ptr(z), 0, 4, 4, 1, None, None, _lib.ptr(out)) != 0
    assert "dimension" in _lib.last_error()
    assert lib.fr3d_get_displacement(C.byref(p), None, _lib.ptr(z), 4, 4, 4, 1, None, None, _lib.ptr(out)) != 0
    assert lib.fr3d_get_displacement(C.byref(p), _lib.ptr(z), _lib.ptr(z), 4, 4, 4, 9, None, None, _lib.ptr(out)) != 0
    p.update_lag = 0
    assert lib.fr3d_get_displacement(C.byref(p), _lib.ptr(z), _lib.ptr(z), 4, 4, 4, 1, None, None, _lib.ptr(out)) != 0
    assert lib.fr3d_warp(_lib.ptr(z), 0, _lib.ptr(out), 0, _lib.ptr(z), 4, 4, 4, 1, 2, _lib.ptr(z)) != 0
    assert _lib.last_error().startswith("Unsupported interpolation")


def test_stream_probe_reports_a_plausible_rate(hip):
    import ctypes as C
    from flowreg3d_amd import _lib
    lib = _lib.init()
    rate = C.c_double(0.0)
    _lib.check(lib.fr3d_stream_probe(1 << 26, 5, C.byref(rate)))
    assert 500.0 < rate.value < 8000.0, rate.value  # GB/s: below the nominal HBM peak, far above PCIe
    assert lib.fr3d_stream_probe(0, 5, C.byref(rate)) != 0
    read = C.c_double(0.0)
    _lib.check(lib.fr3d_read_probe(1 << 24, 5, C.byref(read)))
    assert 500.0 < read.value < 8000.0, read.value
