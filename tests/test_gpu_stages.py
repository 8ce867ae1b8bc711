"""-m gpu: every HIP stage of the hot path against the CPU oracle and the reference's golden
vectors, through the C ABI (flowreg3d_amd.core -> ctypes -> libflowreg3d_hip.so)."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


# ---- K1 resample: bit-exact vs oracle (same table code, fp32 sequential accumulation) -----------
@pytest.mark.parametrize("shape,size", [((18, 22, 26), (11, 15, 17)), ((18, 22, 26), (23, 28, 33)),
                                        ((18, 22, 26), (18, 30, 13)), ((40, 70, 90), (13, 23, 30)),
                                        ((5, 6, 7), (5, 6, 7)), ((1, 9, 9), (1, 5, 12)),
                                        # every axis longer: the one-kernel form of the three passes (LDS tiles), several
                                        # tiles per axis, reflection at both ends, scale 1.25 (pyramid step) up to 22x
                                        ((41, 52, 105), (52, 65, 131)), ((30, 33, 70), (75, 70, 150)),
                                        ((9, 9, 9), (40, 41, 200)), ((20, 21, 22), (20, 27, 28))])
def test_resize_bit_exact_vs_oracle(hip, oracle, shape, size):
    rng = np.random.default_rng(1)
    vol = rng.random(shape, dtype=np.float32)
    got = hip.imresize_fused_gauss_cubic3D(vol, size)
    want = oracle.imresize_fused_gauss_cubic3D(vol, size)
    assert got.dtype == np.float32 and got.shape == tuple(size)
    assert np.array_equal(got, want)


def test_resize_vs_reference_golden(hip):
    g = golden("k1_resize")
    for name, size in (("down", (11, 15, 17)), ("up", (23, 28, 33)), ("mixed", (18, 30, 13))):
        got = hip.imresize_fused_gauss_cubic3D(g["vol"], size)
        # reference tables use NumPy's SIMD expf (<= 1 ulp from libm): few-ulp agreement
        assert np.abs(got - g[name]).max() < 5e-7
    got4 = hip.imresize_fused_gauss_cubic3D(g["vol4"], (12, 14, 20))
    assert got4.dtype == np.float64 and np.abs(got4 - g["down4"]).max() < 5e-7


# ---- K2 warp ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("method", ["cubic", "linear"])
def test_warp_vs_reference_golden(hip, method):
    g = golden("k2_warp")
    got = hip.imregister_wrapper(g["f2"], g["u"], g["v"], g["w"], g["f1"], method)
    want = g[method]
    assert got.dtype == np.float32 and got.shape == want.shape
    # fp64 coefficients and weights like SciPy; output rounded to fp32 -> at most 1 ulp apart
    assert np.abs(got - want).max() <= 1.2e-7
    assert (got != want).mean() < 0.01


def test_warp_single_channel_and_f32_inputs(hip, oracle):
    g = golden("k2_warp")
    got = hip.imregister_wrapper(g["f2"][..., 0], g["u"], g["v"], g["w"], g["f1"][..., 0])
    assert got.shape == g["cubic_c1"].shape and np.abs(got - g["cubic_c1"]).max() <= 1.2e-7
    f2 = g["f2"].astype(np.float32)
    f1 = g["f1"].astype(np.float32)
    u, v, w = (g[k].astype(np.float32) for k in "uvw")
    got = hip.imregister_wrapper(f2, u, v, w, f1, "cubic")
    want = oracle.imregister_wrapper(f2, u, v, w, f1, "cubic")
    assert np.abs(got - want).max() <= 1.2e-7


def test_warp_long_axis_tiled_prefilter(hip, oracle):
    # X + 24 > 64 exercises the LDS-tiled x-axis prefilter; short Z exercises the faithful init
    rng = np.random.default_rng(5)
    f2 = rng.random((7, 21, 150), dtype=np.float32)
    f1 = rng.random((7, 21, 150), dtype=np.float32)
    u, v, w = ((rng.random((7, 21, 150), dtype=np.float32) - 0.5) * 5 for _ in range(3))
    got = hip.imregister_wrapper(f2, u, v, w, f1)
    want = oracle.imregister_wrapper(f2, u, v, w, f1)
    assert np.abs(got - want).max() <= 1.2e-7


@pytest.mark.parametrize("shape,dtype", [((41, 47, 70), np.float32), ((44, 41, 131), np.float64), ((52, 66, 43), np.uint16)])
def test_warp_pad_free_prefilter_vs_oracle(hip, oracle, shape, dtype):
    """Every axis >= 41: the prefilter keeps the 12 pad samples in registers and stores coefficients -2 .. N+1 only
    (k_warp.hip 2b).  Displacements up to 3 voxels, so edge voxels sample right at and beyond the border; two channels
    (interleaved source) and a raw integer volume."""
    rng = np.random.default_rng(11)
    C = 2
    if np.issubdtype(dtype, np.integer):
        f2 = rng.integers(0, 60000, shape + (C,)).astype(dtype)
        f1 = rng.integers(0, 60000, shape + (C,)).astype(np.float32)
    else:
        f2 = rng.random(shape + (C,)).astype(dtype)
        f1 = rng.random(shape + (C,)).astype(dtype)
    u, v, w = (rng.uniform(-3, 3, shape).astype(np.float32) for _ in range(3))
    got = hip.imregister_wrapper(f2.astype(np.float32) if dtype == np.uint16 else f2, u, v, w, f1)
    want = oracle.imregister_wrapper(f2.astype(np.float32) if dtype == np.uint16 else f2, u, v, w, f1)
    scale = 60000.0 if dtype == np.uint16 else 1.0
    assert np.abs(got - want).max() <= 1.2e-7 * scale * 4
    assert (got != want).mean() < 0.01


def test_pad_free_prefilter_is_bit_identical_to_the_padded_form(hip):
    """Same warp with FR3D_PREFILTER=padded (SciPy's layout: 12 stored pad voxels per side; the switch exists in the
    experiment build of the library only) in a child process, against the shipped library."""
    import os, subprocess, sys
    code = ("import numpy as np, hashlib, flowreg3d_amd as fr\n"
            "rng = np.random.default_rng(4)\n"
            "shape = (45, 58, 77)\n"
            "f2 = rng.random(shape + (2,)).astype(np.float32); f1 = rng.random(shape + (2,)).astype(np.float32)\n"
            "u, v, w = (rng.uniform(-4, 4, shape).astype(np.float32) for _ in range(3))\n"
            "f2[:9] = 0.0\n"
            "out = fr.imregister_wrapper(f2, u, v, w, f1)\n"
            "from flowreg3d_amd.executor import HipExecutor3D\n"
            "raw = rng.integers(0, 65535, (2,) + shape + (1,)).astype(np.uint16)\n"
            "proc = (raw / 65535.0).astype(np.float32)\n"
            "reg, fl = HipExecutor3D().process_batch(raw, proc, proc[0].astype(np.float64), proc[0], np.zeros(shape + (3,), np.float32),\n"
            "    flow_params=dict(alpha=(1.0, 1.0, 1.0), iterations=4, levels=2, a_smooth=1.0))\n"
            "assert reg.dtype == np.uint16\n"
            "h = hashlib.sha256(np.ascontiguousarray(out).tobytes()); h.update(np.ascontiguousarray(reg).tobytes())\n"
            "print('HASH', h.hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for mode in ("compact", "padded"):
        env = dict(os.environ, FR3D_PREFILTER=mode, PYTHONPATH=root)
        if mode == "padded":
            env["FR3D_LIB"] = os.path.join(root, "flowreg3d_amd", "lib", "libflowreg3d_hip_exp.so")
        r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[mode] = [l for l in r.stdout.splitlines() if l.startswith("HASH")][0]
    assert out["compact"] == out["padded"]


def test_warp_identity_and_oob(hip):
    rng = np.random.default_rng(2)
    vol = rng.random((9, 10, 11), dtype=np.float32)
    ref = rng.random((9, 10, 11), dtype=np.float32)
    z = np.zeros((9, 10, 11), np.float32)
    assert np.abs(hip.imregister_wrapper(vol, z, z, z, ref) - vol).max() < 1e-6
    big = np.full((9, 10, 11), 100.0, np.float32)
    assert np.array_equal(hip.imregister_wrapper(vol, big, z, z, ref), ref)  # all out of bounds
    with pytest.raises(ValueError):
        hip.imregister_wrapper(vol, z, z, z, ref, "nearest")


# ---- K3 motion tensor -------------------------------------------------------------------------------
def test_motion_tensor_vs_reference_golden(hip):
    g = golden("k3_tensor")
    f1 = g["f1"].astype(np.float32)
    f2 = g["f2"].astype(np.float32)
    assert np.array_equal(f1, g["f1"]) and np.array_equal(f2, g["f2"])  # fixtures are fp32-exact
    J = hip.get_motion_tensor_gc(f1, f2, *g["h"])
    for a in range(10):
        want = g["J"][a]
        assert J[a].shape == want.shape
        # fp64 arithmetic in reference order, one rounding to fp32 storage
        assert np.array_equal(J[a], want.astype(np.float32).astype(np.float64))


@pytest.mark.parametrize("h", [(1.0, 1.0, 1.0), (1.25, 1.25, 1.25), (256 / 205, 256 / 164, 512 / 210), (3.7, 0.9, 11.0)])
def test_motion_tensor_bit_exact_vs_oracle_spacings(hip, oracle, h):
    """The kernel divides by the level constants 2h and h^2 through a reciprocal product with two residual
    corrections (div_by_const): it must round like the true division for any spacing, including the exact
    x0.5 / x1 shortcut of the full-resolution level."""
    from flowreg3d_amd.synthetic import make_pair
    f1, f2, _ = make_pair((14, 23, 37), seed=21, scale=0.4)
    f1[2, 3, 4] = f2[2, 3, 4] = 0.0
    f1[:, :2] = 0.0                      # flat regions: zero numerators
    J = hip.get_motion_tensor_gc(f1, f2, *h)
    want = oracle.get_motion_tensor_gc(f1, f2, *h)
    for a in range(10):
        assert np.array_equal(J[a], np.asarray(want[a]).astype(np.float32).astype(np.float64)), a


# ---- K4-K7 solver -------------------------------------------------------------------------------------
def _solve(hip, g, J, wt, it, lag, ad, fp64):
    hx, hy, hz = g["h"]
    return hip.level_solver(*list(J), wt, g["u"], g["v"], g["w"], (0.25, 0.3, 0.35), it, lag, 0,
                            np.array(ad), 1.0, hx, hy, hz, solver_fp64=fp64)


@pytest.mark.parametrize("fp64", [False, True])
@pytest.mark.parametrize("case,it,lag,ad", [("c1_a045_s1", 12, 5, [0.45]), ("c1_a1_s1", 7, 3, [1.0]),
                                            ("c2_a045_s1", 10, 5, [0.45, 0.6])])
def test_level_solver_vs_reference_golden(hip, case, it, lag, ad, fp64):
    g = golden("k7_solver")
    tag = "c1" if case.startswith("c1") else "c2"
    du, dv, dw = _solve(hip, g, g["J_" + tag], g["wt_" + tag], it, lag, ad, fp64)
    want = g[case]
    inner = (slice(1, -1),) * 3
    err = max(np.abs(d[inner] - want[..., k][inner]).max() for k, d in enumerate((du, dv, dw)))
    scale = np.abs(want).max()
    # lexicographic-exact ordering; differences are fp32 storage / arithmetic only
    assert err < (2e-5 if fp64 else 5e-5) * max(scale, 1.0), err


def test_level_solver_a_smooth_vs_reference_golden(hip):
    g = golden("k7_solver")
    hx, hy, hz = g["h"]
    du, dv, dw = hip.level_solver(*list(g["J_c1"]), g["wt_c1"], g["u"], g["v"], g["w"], (0.25, 0.3, 0.35), 6, 2, 0,
                                  np.array([0.45]), 0.5, hx, hy, hz)
    want = g["c1_a045_s05"]
    inner = (slice(1, -1),) * 3
    err = max(np.abs(d[inner] - want[..., k][inner]).max() for k, d in enumerate((du, dv, dw)))
    assert err < 5e-5 * max(np.abs(want).max(), 1.0), err


def test_level_solver_matches_oracle_bigger(hip, oracle):
    # a larger level, 40 iterations: wavefront pipelining of many in-flight iterations
    from flowreg3d_amd.synthetic import make_pair
    f1, f2, _ = make_pair((20, 30, 34), seed=11, scale=0.3)
    hz = hy = hx = 1.0
    J = oracle.get_motion_tensor_gc(f1, f2, hz, hy, hx)
    J4 = [j[..., None] for j in J]
    wt = np.pad(np.ones((20, 30, 34, 1)), ((1, 1), (1, 1), (1, 1), (0, 0)))
    z = np.zeros((22, 32, 36))
    want = oracle.compute_flow_3d(*J4, wt, z, z, z, 0.25, 0.25, 0.25, 40, 5, np.array([0.45]), 1.0, hx, hy, hz)
    du, dv, dw = hip.level_solver(*J4, wt, z, z, z, (0.25, 0.25, 0.25), 40, 5, 0, np.array([0.45]), 1.0,
                                  hx, hy, hz)
    inner = (slice(1, -1),) * 3
    got = np.stack([du[inner], dv[inner], dw[inner]], -1)
    epe = np.linalg.norm(got - want[inner], axis=-1)
    assert epe.mean() < 2e-5 and epe.max() < 2e-3, (epe.mean(), epe.max())


# ---- K8 median ---------------------------------------------------------------------------------------
def test_median_vs_reference_golden(hip):
    g = golden("k8_median")
    for k in ("a", "b"):
        x = g[k].astype(np.float32)
        got = hip.median_filter5(x)
        # the fp32 cast is monotone, so the median of the casts is the cast of the median
        assert np.array_equal(got, g[k + "_med"].astype(np.float32))


@pytest.mark.parametrize("shape", [(6, 6, 6), (7, 33, 65), (12, 9, 130), (6, 7, 8), (9, 6, 9), (6, 7, 7),
                                   (6, 11, 517), (6, 10, 1030), (40, 41, 43)])
def test_median_bit_exact_vs_oracle(hip, oracle, shape):
    """Row tiles (X < 8) and the flattened kernel: rows that start, end and break anywhere inside a workgroup, odd and
    even widths, rows longer than one workgroup (its neighbour columns come from the extra sorting pass)."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape).astype(np.float32)
    x[1, 2, 3] = x[2, 2, 2]  # ties
    assert np.array_equal(hip.median_filter5(x), oracle.median5(x).astype(np.float32))


def test_fused_median_update_is_bit_identical_to_separate_launches(hip):
    """The engine's default tail of a level (one launch: medians of du, dv, dw added to the flow) against the
    separate per-field median + accumulate launches (FR3D_MEDIAN=3, a switch of the experiment build of the library,
    read at its first use, hence the child process): the whole get_displacement result must agree bit for bit."""
    import os, subprocess, sys, hashlib
    code = ("import numpy as np, hashlib, flowreg3d_amd as fr\n"
            "from flowreg3d_amd import synthetic\n"
            "f, m, _ = synthetic.make_pair((24, 45, 51), seed=5)\n"
            "w = fr.get_displacement(f, m, alpha=(0.25, 0.25, 0.25), iterations=12, update_lag=5, levels=3, a_smooth=1.0)\n"
            "print('HASH', hashlib.sha256(np.ascontiguousarray(w).tobytes()).hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for mode in ("0", "3"):
        env = dict(os.environ, FR3D_MEDIAN=mode, PYTHONPATH=root)
        if mode != "0":
            env["FR3D_LIB"] = os.path.join(root, "flowreg3d_amd", "lib", "libflowreg3d_hip_exp.so")
        r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[mode] = [l for l in r.stdout.splitlines() if l.startswith("HASH")][0]
    assert out["0"] == out["3"]


def test_warp_survives_nan_and_huge_displacements(hip):
    """The gather takes its unclamped fast path only when all 64 taps lie inside the padded grid; NaN, +-inf
    and far-out-of-range displacements must neither fault nor change in-range voxels."""
    rng = np.random.default_rng(3)
    shape = (6, 9, 11)
    vol = rng.random(shape).astype(np.float32)
    ref = rng.random(shape).astype(np.float32)
    u = np.zeros(shape, np.float32)
    v = np.zeros(shape, np.float32)
    w = np.zeros(shape, np.float32)
    u[0, 0, 0], v[1, 1, 1], w[2, 2, 2], u[3, 3, 3], v[4, 4, 4] = np.nan, np.inf, -np.inf, 1e30, -3e9
    got = np.asarray(hip.imregister_wrapper(vol, u, v, w, ref))
    assert got.shape == shape
    mask = np.ones(shape, bool)
    for p in ((0, 0, 0), (1, 1, 1), (2, 2, 2), (3, 3, 3), (4, 4, 4)):
        mask[p] = False
    assert np.allclose(got[mask], vol[mask], atol=1e-6)          # zero displacement: the volume itself
    for p in ((1, 1, 1), (2, 2, 2), (3, 3, 3), (4, 4, 4)):
        assert got[p] == ref[p]                                   # out of bounds -> reference value


def test_resampler_options_match_the_reference(hip):
    """imresize_fused_gauss_cubic3D with per_axis / sigma_coeff / integer images (fr3d_resize3d_ex) against the
    reference's own outputs (tests/golden/k1_resize_opts.npz): bit-exact, like the flow path's form."""
    from conftest import golden
    g = golden("k1_resize_opts")
    vol = g["vol"]
    assert np.array_equal(hip.imresize_fused_gauss_cubic3D(vol, (11, 22, 30), per_axis=True), g["per_axis"])
    assert np.array_equal(hip.imresize_fused_gauss_cubic3D(vol, (9, 15, 13), sigma_coeff=0.9, per_axis=True), g["per_axis_s09"])
    assert np.array_equal(hip.imresize_fused_gauss_cubic3D(vol, (11, 15, 17), sigma_coeff=0.3), g["s03"])
    for key, src, size, kw in (("u16_down", "u16", (11, 15, 17), {}), ("u16_up", "u16", (23, 28, 33), {}),
                               ("i16_mixed", "i16", (18, 30, 13), dict(per_axis=True))):
        got = hip.imresize_fused_gauss_cubic3D(g[src], size, **kw)
        assert got.dtype == g[key].dtype and np.array_equal(got, g[key]), key


def test_level_solver_accepts_the_engines_own_fp32_grade_tensor(hip):
    """`level_solver(*get_motion_tensor_gc(...))` -- the tensor comes back as float64 arrays holding the
    engine's float32 storage, whose smallest eigenvalue sits at ~1e-7 of the trace; the rank-3 guard of
    core.tensor_factors scales its threshold to the input's precision instead of refusing it."""
    from flowreg3d_amd.synthetic import make_pair
    fixed, moving, _ = make_pair((12, 20, 18), seed=6)
    J = hip.get_motion_tensor_gc(fixed, moving, 1.0, 1.0, 1.0)
    P, M, N = J[0].shape
    wt = np.zeros((P, M, N))
    wt[1:-1, 1:-1, 1:-1] = 1.0
    z = np.zeros((P, M, N))
    du, dv, dw = hip.level_solver(*J, wt, z, z, z, (0.25, 0.25, 0.25), 10, 5, False, 0.45, 1.0, 1.0, 1.0, 1.0)
    assert du.shape == (P, M, N) and np.isfinite(du).all() and np.abs(du).max() > 0


@pytest.mark.parametrize("channels,a_smooth,lag", [(1, 1.0, 5), (2, 1.0, 2), (1, 0.5, 3), (2, 0.7, 1)])
def test_level_solver_takes_any_tensor_like_the_reference(hip, oracle, channels, a_smooth, lag):
    """VERDICT r3 (boundary restrictions): the reference's level_solver (core/optical_flow_3d.py:262-316) solves whatever
    tensor it is handed.  A tensor that is NOT the rank-3 gradient-constancy tensor has no square-root factors; the
    mirror then solves on the entries in the reference's arithmetic (fr3d_level_solve_tensor): bit-identical to the
    oracle built with the same portable pow, for a_smooth == 1 and != 1, fp64 flow that is not fp32-representable."""
    from flowreg3d_amd.synthetic import make_pair
    from flowreg3d_amd.core import tensor_factors, TensorRankError
    shape = (14, 22, 19)
    fixed, moving, gt = make_pair(shape, seed=6, channels=channels, cheap=True)
    if channels == 1:
        fixed, moving = fixed[..., None], moving[..., None]
    Js = [oracle.get_motion_tensor_gc(fixed[..., c], moving[..., c], 1.0, 1.2, 0.9) for c in range(channels)]
    J = [np.stack([Js[c][q] for c in range(channels)], -1) for q in range(10)]
    rng = np.random.default_rng(5)
    # a brightness-constancy-like rank-1 term on top: rank 4, still symmetric PSD
    g = rng.standard_normal((4,) + J[0].shape) * 0.3
    g[:, 0], g[:, -1], g[:, :, 0], g[:, :, -1], g[:, :, :, 0], g[:, :, :, -1] = 0, 0, 0, 0, 0, 0
    pairs = [(0, 0), (1, 1), (2, 2), (3, 3), (0, 1), (0, 2), (1, 2), (0, 3), (1, 3), (2, 3)]
    J = [J[q] + g[r] * g[c] for q, (r, c) in enumerate(pairs)]
    with pytest.raises(TensorRankError):
        tensor_factors(*[np.moveaxis(j[1:-1, 1:-1, 1:-1], -1, 0) for j in J])
    P, M, N = J[0].shape[:3]
    wt = np.zeros((P, M, N, channels))
    wt[1:-1, 1:-1, 1:-1] = rng.uniform(0.3, 1.0, (P - 2, M - 2, N - 2, channels)).astype(np.float32)
    uvw = [np.pad(0.7 * gt[..., d] + 0.01 * rng.standard_normal(shape), 1, mode="edge") for d in range(3)]  # float64
    a_data = [0.45, 0.6][:channels]
    alpha = (0.3, 0.25, 0.2)
    inner = (slice(1, -1),) * 3

    def check(J_, uvw_):
        got = hip.level_solver(*J_, wt, *uvw_, alpha, 11, lag, False, a_data, a_smooth, 1.0, 1.2, 0.9)
        try:
            oracle.use_build("ppow")
            want = oracle.compute_flow_3d(*J_, wt, *uvw_, *alpha, 11, lag, a_data, a_smooth, 1.0, 1.2, 0.9)
        finally:
            oracle.use_build("")
        for d in range(3):
            assert got[d].dtype == np.float64 and got[d].shape == (P, M, N)
            assert np.array_equal(got[d][inner], want[inner + (d,)]), (d, np.abs(got[d][inner] - want[inner + (d,)]).max())
        return got

    check(J, uvw)
    # ... and a ghost ring of u, v, w that is NOT the edge pad of the interior (the reference reads whatever is there:
    # it enters the surface voxels' stencil and, with a_smooth != 1, psi_smooth): same path, same bits -- with the rank-4
    # tensor and with the plain gradient-constancy tensor
    ring = [a.copy() for a in uvw]
    for a in ring:
        noise = 0.2 * rng.standard_normal(a.shape)
        mask = np.ones(a.shape, bool)
        mask[inner] = False
        a[mask] += noise[mask]
    g_ring = check(J, ring)
    g_edge = hip.level_solver(*J, wt, *uvw, alpha, 11, lag, False, a_data, a_smooth, 1.0, 1.2, 0.9)
    assert not np.array_equal(g_ring[0][inner], g_edge[0][inner]), "the ghost ring must matter"
    Jgc = [np.stack([Js[c][q] for c in range(channels)], -1) for q in range(10)]
    check(Jgc, ring)


def test_workgroups_with_equal_id_mod_8_share_an_xcd(hip):
    """The placement the sweep's XCD-aware tile order relies on for SPEED (never for correctness: results are bit-identical
    under any placement): workgroups are dealt round-robin over the 8 XCDs, so blockIdx.x and blockIdx.x + 8 share one.
    An observation of this hardware and runtime, not a HIP guarantee -- if it stops holding, revisit sor_xcd_group()."""
    from flowreg3d_amd import _lib
    lib = _lib.init(0)
    for gx, gy in ((4096, 1), (4099, 4)):
        a = np.full((gy, gx), -1, np.int32)
        _lib.check(lib.fr3d_xcd_probe(gx, gy, a.ctypes.data))
        assert a.min() >= 0 and a.max() <= 7
        assert np.mean(a[:, 8:] == a[:, :-8]) > 0.98, (gx, gy)
        assert np.bincount(a.reshape(-1), minlength=8).min() > 0.8 * a.size / 8
