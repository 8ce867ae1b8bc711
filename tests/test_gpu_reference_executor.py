"""-m gpu: HipExecutor3D.process_batch against the REFERENCE's own SequentialExecutor3D.process_batch
(motion_correction/parallelization/sequential_3d.py:37-175), recorded in tests/golden/ex_seq.npz by
tools/gen_golden.py with the reference's get_displacement / imregister_wrapper injected and the
pipeline's flow_params dict (compensate_recording_3D.py:301-315).  Cases: one channel float32 raw
(cubic), two channels uint16 raw (cubic), one channel float64 raw with min_level 1 (linear).

Tolerance on `registered`: the reference's own cross-executor bar, rtol 1e-5 / atol 1e-6
(tests/motion_correction/test_parallelization.py:192-198), reached in the parity solver mode
(fp64 solver storage).  The default mode (fp32 solver storage for one channel) moves the flow by
~1e-5 voxels and the registered intensities (camera counts, 200..3200) accordingly; its bound is
stated separately.  Integer raw volumes: SciPy rounds the interpolated value into the raw dtype, so
a 1e-6-voxel flow difference can flip a value that sits at x.5 -- at most one count, on a handful of
voxels."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _case(g, name):
    p = g[f"{name}_params"]
    fp = dict(alpha=tuple(float(x) for x in p[:3]), weight=g[f"{name}_weight"], levels=int(p[6]), min_level=int(p[5]),
              eta=float(p[7]), update_lag=int(p[3]), iterations=int(p[4]), a_smooth=float(p[8]), a_data=float(p[9]))
    method = "cubic" if int(p[10]) == 3 else "linear"
    return (g[f"{name}_batch"], g[f"{name}_batch_proc"], g[f"{name}_ref_raw"], g[f"{name}_ref_proc"], g[f"{name}_w_init"],
            fp, method, g[f"{name}_registered"], g[f"{name}_flows"])


@pytest.mark.parametrize("name", ["c1_f32", "c2_u16", "c1_f64_lin"])
@pytest.mark.parametrize("mode", ["parity", "default"])
def test_hip_executor_matches_reference_sequential_executor(hip, name, mode):
    from flowreg3d_amd.executor import HipExecutor3D
    g = golden("ex_seq")
    batch, bproc, ref_raw, ref_proc, w_init, fp, method, reg_ref, flows_ref = _case(g, name)
    if mode == "parity":
        fp = dict(fp, solver_fp64=2)
    calls = []
    with HipExecutor3D() as ex:
        reg, flows = ex.process_batch(batch, bproc, ref_raw, ref_proc, w_init, None, None, interpolation_method=method,
                                      progress_callback=calls.append, flow_params=fp)
    assert reg.dtype == batch.dtype == reg_ref.dtype and reg.shape == reg_ref.shape
    assert flows.dtype == np.float32 and flows.shape == flows_ref.shape and sum(calls) == batch.shape[0]
    epe = np.linalg.norm(flows.astype(np.float64) - flows_ref.astype(np.float64), axis=-1)
    d = np.abs(reg.astype(np.float64) - reg_ref.astype(np.float64))
    info = f"{name}/{mode}: flow EPE mean {epe.mean():.2e} max {epe.max():.2e}; registered |diff| max {d.max():.3e}"
    print(info)
    from driver_cases import DEFAULT_REG_REL, record_registered
    record_registered(f"executor/{name}/{mode}", d, reg_ref, epe)
    if mode == "parity":
        assert epe.mean() < 1e-5, info
        if np.issubdtype(batch.dtype, np.integer):
            assert d.max() <= 1 and (d > 0).mean() < 1e-3, info
        else:
            np.testing.assert_allclose(reg, reg_ref, rtol=1e-5, atol=1e-6, err_msg=info)
    else:
        # default solver mode (what bench.py times): north-star flow bound, registered within 2e-5 relative
        # of the intensity range for float volumes, one count for integer volumes
        assert epe.mean() < 1e-4, info
        if np.issubdtype(batch.dtype, np.integer):
            assert d.max() <= 1 and (d > 0).mean() < 1e-2, info
        else:
            assert d.max() <= DEFAULT_REG_REL * float(np.ptp(reg_ref)), info
            assert d.mean() < 1e-5 * float(np.abs(reg_ref).max()), info


def test_integer_raw_volumes_are_rounded_like_scipy_not_truncated(hip, oracle):
    """uint8 / uint16 / int16 raw volumes: the final warp rounds half up (unsigned) / half away from zero
    (signed) and saturates, exactly like map_coordinates writing into an integer array; out-of-bounds voxels
    take the float64 reference value through float32 and NumPy's truncating cast."""
    from flowreg3d_amd.executor import HipExecutor3D
    rng = np.random.default_rng(3)
    shape = (9, 14, 12)
    from scipy.ndimage import gaussian_filter
    base = gaussian_filter(rng.random(shape), 1.2)
    base = (base - base.min()) / (base.max() - base.min())
    fp = dict(alpha=(0.25,) * 3, update_lag=5, iterations=6, min_level=0, levels=2, eta=0.8, a_smooth=1.0, a_data=0.45)
    for dt, scale, off in ((np.uint8, 250.0, 3.0), (np.uint16, 65000.0, 300.0), (np.int16, 60000.0, -30000.0)):
        raw = (base * scale + off).astype(dt)[None, ..., None]
        ref_raw = (np.roll(base, 1, axis=2) * scale + off)[..., None]  # float64, like the pipeline's reference
        proc = base[None, ..., None]
        w_init = np.zeros(shape + (3,), np.float32)
        w_init[..., 0] = 2.6  # pushes voxels out of bounds on one face
        for method in ("cubic", "linear"):
            reg, flows = HipExecutor3D().process_batch(raw, proc, ref_raw, np.roll(base, 1, axis=2)[..., None], w_init,
                                                       None, None, interpolation_method=method, flow_params=fp)
            want = oracle.register_raw(raw[0], flows[0], ref_raw, method)
            assert reg.dtype == dt
            d = np.abs(reg[0].astype(np.int64) - want.astype(np.int64))
            # same flow on both sides: identical except where fp64 summation order puts a value on x.5
            assert d.max() <= 1 and (d > 0).mean() < 2e-3, (dt, method, d.max(), (d > 0).mean())
