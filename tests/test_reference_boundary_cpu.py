"""CPU: the boundary pinned to the reference itself.

  * the CPU oracle, driven through the executor body, reproduces the reference's own
    SequentialExecutor3D.process_batch outputs (tests/golden/ex_seq.npz, written by tools/gen_golden.py
    from /root/reference) -- flows to ~1e-6 voxels, registered volumes at the reference's cross-executor
    tolerance (rtol 1e-5 / atol 1e-6, tests/motion_correction/test_parallelization.py:192-198), integer
    raw volumes exactly;
  * HipExecutor3D registers into the REAL flowreg3d._runtime.RuntimeContext (_runtime.py:149-199) and
    resolves back under the name the pipeline asks for ("hip3d", compensate_recording_3D.py:88-94).
    Needs the reference tree (/root/reference, build container only); skipped where it is absent.
"""
import os
import sys
import types

import numpy as np
import pytest

from conftest import golden

REF_SRC = "/root/reference/src"


@pytest.mark.parametrize("name", ["c1_f32", "c2_u16", "c1_f64_lin"])
def test_oracle_executor_body_matches_reference_executor(oracle, name):
    g = golden("ex_seq")
    p = g[f"{name}_params"]
    fp = dict(alpha=tuple(float(x) for x in p[:3]), weight=g[f"{name}_weight"], levels=int(p[6]), min_level=int(p[5]),
              eta=float(p[7]), update_lag=int(p[3]), iterations=int(p[4]), a_smooth=float(p[8]), a_data=float(p[9]))
    method = "cubic" if int(p[10]) == 3 else "linear"
    batch, bproc, ref_raw, ref_proc, w_init = (g[f"{name}_{k}"] for k in ("batch", "batch_proc", "ref_raw", "ref_proc",
                                                                          "w_init"))
    for t in range(batch.shape[0]):
        # parallelization/sequential_3d.py:148-170
        flow = oracle.get_displacement(ref_proc, bproc[t], uvw=w_init.copy(), **fp).astype(np.float32)
        reg = oracle.register_raw(batch[t], flow, ref_raw, method)
        epe = np.linalg.norm(flow.astype(np.float64) - g[f"{name}_flows"][t], axis=-1)
        assert epe.mean() < 5e-6 and epe.max() < 2e-3, (epe.mean(), epe.max())
        want = g[f"{name}_registered"][t]
        assert reg.dtype == want.dtype
        if np.issubdtype(want.dtype, np.integer):
            d = np.abs(reg.astype(np.int64) - want.astype(np.int64))
            assert d.max() <= 1 and (d > 0).mean() < 1e-3
        else:
            np.testing.assert_allclose(reg, want, rtol=1e-5, atol=1e-6)


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference tree not present (GPU box)")
def test_hip_executor_registers_into_the_real_runtime_context():
    if "numba" not in sys.modules:  # the reference imports numba.njit at module level; not installed here
        m = types.ModuleType("numba")
        m.njit = lambda *a, **k: (a[0] if len(a) == 1 and callable(a[0]) and not k else (lambda f: f))
        sys.modules["numba"] = m
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF_SRC)
    try:
        from flowreg3d._runtime import RuntimeContext
        import flowreg3d.motion_correction.parallelization  # noqa: F401  (the reference's executors self-register)
        from flowreg3d_amd import executor
        from flowreg3d_amd.executor import HipExecutor3D
        assert executor.runtime_context() is RuntimeContext
        assert HipExecutor3D.register(force=True) is True
        reg = RuntimeContext.get("parallelization_registry")
        assert reg["hip3d"] == "flowreg3d_amd.executor.HipExecutor3D"
        assert "hip3d" in RuntimeContext.get("available_parallelization")
        # the lookup the pipeline performs (compensate_recording_3D.py:88-121): name + "3d", then cls(n_workers=...)
        cls = RuntimeContext.get_parallelization_executor("hip" + "3d")
        assert cls is HipExecutor3D
        inst = cls(n_workers=4)
        assert inst.name == "hip3d" and inst.n_workers == 1
        info = inst.get_info()
        assert info["name"] == "hip3d" and info["type"] == "HipExecutor3D"
        # the reference's own executors are still there
        assert RuntimeContext.get_parallelization_executor("sequential3d").__name__ == "SequentialExecutor3D"
    finally:
        sys.path.remove(REF_SRC)


@pytest.mark.parametrize("method", ["cubic", "linear"])
def test_oracle_update_reference_matches_reference(oracle, method):
    """f-4: BatchMotionCorrector._update_reference (compensate_recording_3D.py:395-429), golden written by
    tools/gen_golden.py with the reference's imregister_wrapper -- bit-identical."""
    g = golden("f4_update_ref")
    got = oracle.update_reference(g["batch_proc"], g["w"], g["ref_proc"], method)
    assert np.array_equal(got, g[f"new_ref_{method}"])
