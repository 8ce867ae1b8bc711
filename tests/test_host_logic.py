"""CPU: host-side logic of the package (no kernels): schedule, weight handling, parameter packing,
tensor factorisation, executor registry and argument validation."""
import numpy as np
import pytest

from conftest import golden


def test_warping_depth_matches_reference_rows():
    from flowreg3d_amd import warpingDepth
    g = golden("schedule")
    for p, m, n, eta, levels, depth in g["rows"]:
        assert warpingDepth(eta, int(levels), int(p), int(m), int(n)) == int(depth)


@pytest.mark.parametrize("dims,eta,levels,min_level", [((512, 512, 512), 0.8, 5, 0), ((256, 256, 256), 0.8, 4, 0),
                                                       ((32, 64, 64), 0.8, 2, 0), ((256, 512, 512), 0.8, 100, 5),
                                                       ((256, 512, 512), 0.8, 8, 0), ((12, 16, 16), 0.8, 50, 0),
                                                       ((5, 40, 40), 0.8, 3, 0), ((30, 30, 30), 0.5, 10, 7),
                                                       ((20, 28, 28), 0.75, 50, 1)])
def test_engine_schedule_equals_oracle_schedule(oracle, dims, eta, levels, min_level):
    """fr3d_schedule (engine host code) vs the oracle's restatement of optical_flow_3d.py:389-408,
    including Python's round-half-even and the min_level clamp."""
    import __graft_entry__ as g
    g.build()
    from flowreg3d_amd import pyramid_schedule
    assert pyramid_schedule(*dims, eta, levels, min_level) == oracle.schedule(*dims, eta, levels, min_level)


def test_schedule_sizes_of_baseline_configs():
    from flowreg3d_amd import pyramid_schedule
    assert [s[0] for s in pyramid_schedule(256, 256, 256, 0.8, 4)[0]] == [105, 131, 164, 205, 256]
    assert [s[0] for s in pyramid_schedule(512, 512, 512, 0.8, 5)[0]] == [168, 210, 262, 328, 410, 512]
    with pytest.raises(ValueError):
        pyramid_schedule(0, 4, 4, 0.8, 3)


def test_expand_weight_follows_reference_rules(oracle):
    from flowreg3d_amd import expand_weight
    assert expand_weight(None, 2, 3, 4, 2) is None
    for w in (np.array([0.7, 0.3]), np.array([2.0]), np.array([1.0, 2.0, 3.0]), np.ones((2, 3, 4)) * 0.5):
        a = expand_weight(w, 2, 3, 4, 2)
        b = oracle.expand_weight(w, 2, 3, 4, 2)
        assert a.shape == (2, 3, 4, 2) and np.array_equal(a, b)
    w4 = np.random.default_rng(0).random((2, 3, 4, 2))
    assert np.array_equal(expand_weight(w4, 2, 3, 4, 2), w4)
    with pytest.raises(ValueError):
        expand_weight(np.ones((3, 3, 4, 2)), 2, 3, 4, 2)


def test_make_params():
    from flowreg3d_amd import _lib
    p = _lib.make_params((0.25, 0.5, 1.0), 5, 100, 0, 4, 0.8, 1.0, [0.45, 0.6], 2)
    assert list(p.alpha) == [0.25, 0.5, 1.0] and p.iterations == 100 and p.update_lag == 5
    assert p.a_data[0] == 0.45 and p.a_data[1] == 0.6 and p.solver_fp64 == -1  # FR3D_SOLVER_AUTO
    assert _lib.make_params(1, 10, 20, 0, 50, 0.8, 1.0, 0.45, 1, solver_fp64=0).solver_fp64 == 0
    assert _lib.make_params(1, 10, 20, 0, 50, 0.8, 1.0, 0.45, 1, solver_fp64=2).solver_fp64 == 2
    p = _lib.make_params(2, 10, 20, 0, 50, 0.8, 1.0, 0.45, 3, solver_fp64=True)
    assert list(p.alpha) == [2.0, 2.0, 2.0] and p.a_data[2] == 0.45 and p.solver_fp64 == 1
    with pytest.raises(ValueError):
        _lib.make_params((1, 2), 10, 20, 0, 50, 0.8, 1.0, 0.45, 1)
    with pytest.raises(ValueError):
        _lib.make_params(1, 10, 20, 0, 50, 0.8, 1.0, [0.1, 0.2, 0.3], 2)


def test_tensor_factors_reproduce_reference_tensor():
    from flowreg3d_amd import tensor_factors
    g = golden("k3_tensor")
    J = [j[1:-1, 1:-1, 1:-1] for j in g["J"]]
    A = tensor_factors(*J)
    assert A.shape == (12,) + J[0].shape
    a = A.reshape(3, 4, *J[0].shape)
    pairs = {0: (0, 0), 1: (1, 1), 2: (2, 2), 3: (3, 3), 4: (0, 1), 5: (0, 2), 6: (1, 2), 7: (0, 3), 8: (1, 3), 9: (2, 3)}
    for k, (r, c) in pairs.items():
        assert np.abs((a[:, r] * a[:, c]).sum(0) - J[k]).max() < 1e-10 * max(1.0, np.abs(J[k]).max())


def test_argument_validation_happens_before_the_device():
    import flowreg3d_amd as fr
    z = np.zeros((8, 8, 8), np.float32)
    with pytest.raises(ValueError):
        fr.imregister_wrapper(z, z, z, z, z, "spline")  # same message/exception as the reference
    with pytest.raises(ValueError):
        fr.get_displacement(z, np.zeros((8, 8, 9), np.float32), a_smooth=1.0)
    with pytest.raises(ValueError):
        fr.get_displacement(z, z, a_smooth=1.0, uvw=np.zeros((8, 8, 8, 2)))


def test_executor_registry_roundtrip():
    from flowreg3d_amd.executor import HipExecutor3D, _LocalRuntimeContext
    ex = HipExecutor3D(n_workers=7)
    assert ex.name == "hip3d" and ex.n_workers == 1
    info = ex.get_info()
    assert info["name"] == "hip3d" and info["type"] == "HipExecutor3D"
    _LocalRuntimeContext.register_parallelization_executor("hip3d", HipExecutor3D)
    assert _LocalRuntimeContext.get_parallelization_executor("hip3d") is HipExecutor3D
    assert _LocalRuntimeContext.get_parallelization_executor("nope") is None
    assert _LocalRuntimeContext.get_parallelization_executor(None) is None
    _LocalRuntimeContext.register_parallelization_executor("x3d", "flowreg3d_amd.executor.HipExecutor3D")
    assert _LocalRuntimeContext.get_parallelization_executor("x3d") is HipExecutor3D


def test_executor_rejects_unsupported_requests_without_gpu():
    from flowreg3d_amd.executor import HipExecutor3D
    ex = HipExecutor3D()
    b = np.zeros((1, 6, 6, 6, 1), np.float32)
    r = np.zeros((6, 6, 6, 1), np.float32)
    w = np.zeros((6, 6, 6, 3), np.float32)
    with pytest.raises(ValueError):
        ex.process_batch(b, b, r, r, w, None, None, interpolation_method="bogus", flow_params={"a_smooth": 1.0})
    with pytest.raises(NotImplementedError):
        ex.process_batch(b, b, r, r, w, None, None, flow_params={"a_smooth": 1.0, "cc_initialization": True})


def test_shard_indices():
    from flowreg3d_amd.distributed import shard_indices
    assert shard_indices(64, 0, 8) == list(range(0, 64, 8))
    assert sorted(sum((shard_indices(10, r, 4) for r in range(4)), [])) == list(range(10))
    assert shard_indices(2, 3, 4) == []
    with pytest.raises(ValueError):
        shard_indices(4, 4, 4)


def test_pipeline_option_helpers_follow_ofoptions():
    """OF_options_3D.py:239-264 (alpha), :329-341 (effective_min_level), :371-399 (get_weight_at)."""
    from flowreg3d_amd.pipeline import Options, _alpha3, _weight_at
    assert _alpha3(2.0) == (2.0, 2.0, 2.0) and _alpha3((1.0, 2.0)) == (1.0, 2.0, 2.0) and _alpha3((1, 2, 3)) == (1.0, 2.0, 3.0)
    with pytest.raises(ValueError):
        _alpha3((1, 2, 3, 4))
    o = Options()
    assert o.effective_min_level == 5 and o.iterations == 100 and o.update_lag == 5 and o.a_smooth == 1.0
    assert Options(min_level=-1, quality_setting="balanced").effective_min_level == 4
    assert Options(min_level=-1, quality_setting="fast").effective_min_level == 6
    assert Options(min_level=-1).effective_min_level == 0
    assert _weight_at([0.5, 0.5], 0, 1) == 1.0            # truncated and renormalised
    assert _weight_at([0.7, 0.3], 1, 2) == 0.3
    assert _weight_at([0.6], 0, 2) == 0.6 and _weight_at([0.6, 0.4], 2, 3) == pytest.approx(1 / 3)
    w3 = np.arange(2 * 2 * 2 * 2, dtype=float).reshape(2, 2, 2, 2)
    assert np.array_equal(_weight_at(w3, 1, 2), w3[1])


def test_tensor_factors_refuses_a_tensor_that_is_not_rank_three():
    """The device solver rebuilds the motion tensor from three square-root factors; a rank-4 tensor (another
    constancy assumption) must be rejected, not silently solved as a different system."""
    import pytest
    from conftest import golden
    from flowreg3d_amd.core import tensor_factors
    J = golden("k3_tensor")["J"]
    Ji = [J[i][1:-1, 1:-1, 1:-1] for i in range(10)]
    A = tensor_factors(*Ji)
    assert A.shape == (12,) + Ji[0].shape
    # J = sum_k a_k a_k^T reproduces the entries (index 4k + column; J11 = sum_k a_k[0]^2)
    np.testing.assert_allclose(A[0] ** 2 + A[4] ** 2 + A[8] ** 2, Ji[0], rtol=1e-9, atol=1e-12)
    bad = [j.copy() for j in Ji]
    bad[3] = bad[3] + 1.0  # J44 alone: a fourth independent direction
    with pytest.raises(ValueError):
        tensor_factors(*bad)


@pytest.mark.parametrize("dims", [(7, 9, 11), (1, 1, 1), (1, 70, 3), (5, 3, 130), (12, 66, 65), (20, 20, 20), (3, 200, 2)])
@pytest.mark.parametrize("iterations", [12, 1, 7])
def test_sor_chain_schedule_issues_every_update_once(dims, iterations):
    """Host replay of the sweep kernel's index arithmetic (fr3d_sor_schedule_check, no GPU): every voxel update of
    level_solver_3d.py:383-540 is issued exactly once, by launch i+j+k+2t, for every tile shape (rows x chained
    iterations) including the engine's."""
    import ctypes as C
    from flowreg3d_amd import _lib
    lib = _lib.load()
    Z, Y, X = dims
    for by, nch in ((0, 0), (1, 1), (2, 1), (2, 4), (4, 2), (1, 8), (8, 1), (2, 3)):
        n = C.c_longlong(0)
        bad = lib.fr3d_sor_schedule_check(Z, Y, X, iterations, by, nch, C.byref(n))
        assert bad == 0, (dims, iterations, by, nch, bad)
        assert n.value == Z * Y * X * iterations
