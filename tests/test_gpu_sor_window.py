"""The window sweep (k_sor_win.hip: a psi window per workgroup on chip) against the plane-launch sweep (k_sor.hip).

Both evaluate the lexicographic SOR of core/level_solver_3d.py:383-540 with the same per-voxel functions and the same
rounding to the storage format, so whole pyramids must agree BIT FOR BIT in every solver mode -- through every tile
boundary, window boundary, psi period longer or shorter than a window, and channel count the kernel covers.
(The kernel's indexing is also emulated on the CPU: tools/emu/sor_win_emu.hip.)"""
import numpy as np
import pytest

from flowreg3d_amd.synthetic import make_pair

pytestmark = pytest.mark.gpu

CASES = [
    # shape, channels, iterations, update_lag, levels, modes
    ((24, 40, 36), 1, 20, 5, 2, (0, 1, 2, 3)),     # several tiles in y and z, partly filled tiles
    ((33, 35, 50), 1, 12, 5, 3, (1, 2, 3)),        # last window shorter than the lag
    ((40, 20, 33), 1, 14, 7, 2, (2, 3)),           # psi period longer than a window: the system is re-read
    ((20, 48, 24), 1, 9, 3, 2, (1, 2)),            # psi period shorter than the window size
    ((18, 34, 40), 1, 6, 1, 1, (2,)),              # a psi update on every iteration
    ((36, 36, 20), 2, 10, 5, 2, (1, 3)),           # two channels (with fp64 storage they take the plane sweep: LDS)
    ((7, 9, 11), 1, 10, 5, 1, (1, 2)),             # one partly filled tile
]


@pytest.mark.parametrize("shape,channels,iters,lag,levels,modes", CASES)
def test_window_sweep_is_bit_identical_to_plane_sweep(hip, shape, channels, iters, lag, levels, modes):
    fixed, moving, _ = make_pair(shape, seed=11, channels=channels, scale=0.6)
    for mode in modes:
        kw = dict(alpha=(0.25, 0.25, 0.25), update_lag=lag, iterations=iters, levels=levels, eta=0.8, a_smooth=1.0,
                  a_data=0.45, solver_fp64=mode)
        ref = hip.get_displacement(fixed, moving, solver_sweep=1, **kw)
        win = hip.get_displacement(fixed, moving, solver_sweep=2, **kw)
        assert np.isfinite(ref).all()
        assert np.abs(ref).max() > 0.05, "degenerate case"
        assert np.array_equal(ref, win), f"mode {mode}: max diff {np.abs(ref - win).max():.3e}"


def test_window_sweep_in_a_lockstep_batch(hip):
    """volumes of a batch share the launches (blockIdx.y): same bits as single calls"""
    from flowreg3d_amd.executor import HipExecutor3D
    shape = (20, 36, 28)
    ref_v, _, _ = make_pair(shape, seed=5)
    movs = [make_pair(shape, seed=5, scale=s)[1] for s in (0.3, 0.6, 0.9)]
    batch = np.stack(movs)[..., None].astype(np.float32)
    refp = ref_v[..., None].astype(np.float64)
    ex = HipExecutor3D()
    fp = dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=15, levels=2, min_level=0, eta=0.8, a_smooth=1.0,
              a_data=0.45, solver_fp64=2)
    outs = {}
    for sweep in ("planes", "window"):
        _, flows = ex.process_batch(batch, batch.astype(np.float64), refp, refp, np.zeros(shape + (3,), np.float32),
                                    None, None, flow_params=dict(fp, solver_sweep={"planes": 1, "window": 2}[sweep]))
        outs[sweep] = flows
    assert np.array_equal(outs["planes"], outs["window"])
