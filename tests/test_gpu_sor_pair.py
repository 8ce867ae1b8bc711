"""-m gpu: the two-hyperplane sweep (k_sor_pair.hip) reproduces the one-hyperplane sweep (k_sor_step,
k_sor.hip) BIT FOR BIT -- same arithmetic (k_sor_core.h), same lexicographic Gauss-Seidel order
(core/level_solver_3d.py:383-540), only the launch schedule and the path of the increments differ.
Covers odd/even iteration counts (the increments are double-buffered by iteration parity), shapes whose
rows are shorter than, equal to and longer than a 64-lane tile, axes of length 1, 1-3 channels, all
three solver storage/arithmetic modes, psi updates on the first/every/no iteration and lock-step batches."""
import os

import numpy as np
import pytest
from scipy.ndimage import gaussian_filter

pytestmark = pytest.mark.gpu


def _vol(rng, shape):
    a = gaussian_filter(rng.random(shape), 1.0, mode="reflect")
    return ((a - a.min()) / (a.max() - a.min() + 1e-12)).astype(np.float32)


def _run(kernel, fn):
    old = os.environ.get("FR3D_SOR_KERNEL")
    os.environ["FR3D_SOR_KERNEL"] = kernel
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop("FR3D_SOR_KERNEL", None)
        else:
            os.environ["FR3D_SOR_KERNEL"] = old


CASES = [
    # shape, C, iterations, update_lag, levels, mode
    ((20, 41, 41), 1, 12, 5, 2, 1),
    ((9, 11, 13), 1, 1, 1, 1, 0),
    ((9, 11, 13), 2, 2, 3, 1, 2),
    ((33, 70, 65), 1, 7, 2, 3, 1),
    ((64, 64, 64), 1, 10, 5, 2, 0),
    ((5, 130, 7), 1, 5, 5, 1, 1),
    ((70, 3, 129), 3, 4, 2, 2, 2),
    ((1, 40, 40), 1, 6, 4, 2, 1),
    ((17, 1, 23), 1, 3, 7, 1, 1),
    ((1, 1, 1), 1, 3, 1, 1, 1),
    ((40, 100, 200), 1, 9, 5, 3, 1),
    ((30, 30, 30), 1, 0, 5, 2, 1),
]


@pytest.mark.parametrize("shape,C,iters,lag,levels,mode", CASES)
@pytest.mark.parametrize("kernel", ["pair6", "pair14"])
def test_pair_sweep_is_bit_identical_to_step_sweep(hip, kernel, shape, C, iters, lag, levels, mode):
    rng = np.random.default_rng(abs(hash((shape, C, iters))) % (2 ** 31))
    fixed = np.stack([_vol(rng, shape) for _ in range(C)], -1)
    moving = np.stack([0.97 * gaussian_filter(fixed[..., c], 0.6) + 0.02 for c in range(C)], -1).astype(np.float32)
    kw = dict(alpha=(0.25, 0.3, 0.35), update_lag=lag, iterations=iters, min_level=0, levels=levels, eta=0.8,
              a_smooth=1.0, a_data=0.45, solver_fp64=mode)
    want = _run("step", lambda: hip.get_displacement(fixed, moving, **kw))
    got = _run(kernel, lambda: hip.get_displacement(fixed, moving, **kw))
    assert np.isfinite(want).all()
    assert np.array_equal(want, got), float(np.abs(want - got).max())


@pytest.mark.parametrize("kernel", ["pair6", "pair14"])
def test_pair_sweep_in_a_lockstep_batch(hip, kernel):
    """fr3d_process_batch with a lock-step batch of 3: the shared launches of the pair sweep give the same
    flows and registered volumes as the step sweep."""
    from flowreg3d_amd.executor import HipExecutor3D
    rng = np.random.default_rng(5)
    shape = (24, 48, 40)
    ref = _vol(rng, shape)[..., None]
    batch = np.stack([(0.95 * gaussian_filter(ref[..., 0], 0.4 + 0.2 * t) + 0.01 * t)[..., None] for t in range(5)], 0)
    batch = batch.astype(np.float32)
    fp = dict(alpha=(0.25, 0.25, 0.25), weight=np.array([1.0]), levels=3, min_level=0, eta=0.8, update_lag=5,
              iterations=11, a_smooth=1.0, a_data=0.45)
    ex = HipExecutor3D()

    def go():
        hip._lib.load().fr3d_set_batch(3)
        try:
            return ex.process_batch(batch, batch.astype(np.float64), ref.astype(np.float64), ref.astype(np.float64),
                                    np.zeros(shape + (3,), np.float32), flow_params=fp)
        finally:
            hip._lib.load().fr3d_set_batch(0)
    reg0, fl0 = _run("step", go)
    reg1, fl1 = _run(kernel, go)
    assert np.array_equal(fl0, fl1) and np.array_equal(reg0, reg1)
