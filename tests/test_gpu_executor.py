"""-m gpu: the HipExecutor3D plugin (fr3d_process_batch) against the restated sequential executor
body on the CPU oracle -- same contract the reference tests for its executors
(tests/motion_correction/test_parallelization.py:152-198: cross-executor rtol=1e-5, atol=1e-6 on the
registered output)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _series(T=3, shape=(12, 20, 24), C=1, dtype=np.float32):
    from flowreg3d_amd.synthetic import make_pair
    fixed, _, _ = make_pair(shape, seed=21, channels=C)
    vols = [make_pair(shape, seed=21, channels=C, scale=0.25 * (t + 1))[1] for t in range(T)]
    batch = np.stack(vols).astype(np.float32)
    if C == 1:
        batch = batch[..., None]
        fixed = fixed[..., None]
    return fixed.astype(np.float32), batch


def _oracle_batch(oracle, batch, batch_proc, ref_raw, ref_proc, w_init, fp, method):
    T = batch.shape[0]
    reg = np.empty_like(batch)
    flows = np.empty(batch.shape[:4] + (3,), np.float32)
    for t in range(T):
        f = oracle.get_displacement(ref_proc, batch_proc[t], uvw=w_init.copy(), **fp).astype(np.float32)
        r = oracle.imregister_wrapper(batch[t], f[..., 0], f[..., 1], f[..., 2], ref_raw, method)
        flows[t] = f
        reg[t] = r.reshape(reg[t].shape)
    return reg, flows


@pytest.mark.parametrize("C,method", [(1, "cubic"), (2, "cubic"), (1, "linear")])
def test_process_batch_matches_sequential_oracle(hip, oracle, C, method):
    from flowreg3d_amd.executor import HipExecutor3D
    fixed, batch = _series(C=C)
    w0 = np.zeros(batch.shape[1:4] + (3,), np.float32)
    w0[..., 0] = 0.3
    fp = dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=20, min_level=0, levels=3, eta=0.8,
              a_smooth=1.0, a_data=0.45, weight=np.ones(batch.shape[1:]) / C)
    calls = []
    with HipExecutor3D() as ex:
        reg, flows = ex.process_batch(batch, batch.astype(np.float64), fixed, fixed.astype(np.float64), w0, None,
                                      None, interpolation_method=method, progress_callback=calls.append,
                                      flow_params=fp)
    assert reg.shape == batch.shape and reg.dtype == batch.dtype
    assert flows.shape == batch.shape[:4] + (3,) and flows.dtype == np.float32
    assert sum(calls) == batch.shape[0]
    reg_o, flows_o = _oracle_batch(oracle, batch, batch, fixed, fixed, w0, fp, method)
    epe = np.linalg.norm(flows.astype(np.float64) - flows_o, axis=-1)
    assert epe.mean() < 1e-4, (epe.mean(), epe.max())
    # the reference's own cross-executor tolerance is rtol=1e-5, atol=1e-6; the flow differs at the
    # 1e-5 level (fp32 solver storage), which moves warped intensities by ~1e-5 * |grad|
    assert np.abs(reg - reg_o).max() < 5e-4
    assert np.abs(reg - reg_o).mean() < 1e-5


def test_registered_is_cast_to_batch_dtype(hip):
    from flowreg3d_amd.executor import HipExecutor3D
    fixed, batch = _series(T=2)
    raw16 = np.round(batch * 4000).astype(np.uint16)
    ref16 = np.round(fixed * 4000).astype(np.float64)
    w0 = np.zeros(batch.shape[1:4] + (3,), np.float32)
    fp = dict(alpha=(0.25,) * 3, update_lag=5, iterations=10, min_level=0, levels=2, eta=0.8, a_smooth=1.0,
              a_data=0.45)
    reg, flows = HipExecutor3D().process_batch(raw16, batch, ref16, fixed, w0, None, None, flow_params=fp)
    assert reg.dtype == np.uint16 and reg.shape == raw16.shape
    assert np.isfinite(flows).all() and not np.all(flows == 0)


def test_executor_registers_when_gpu_present(hip):
    from flowreg3d_amd.executor import HipExecutor3D, runtime_context
    assert HipExecutor3D.register() is True
    assert runtime_context().get_parallelization_executor("hip3d") is HipExecutor3D


def test_empty_batch(hip):
    from flowreg3d_amd.executor import HipExecutor3D
    fixed, batch = _series(T=1)
    w0 = np.zeros(batch.shape[1:4] + (3,), np.float32)
    fp = dict(alpha=(0.25,) * 3, update_lag=5, iterations=5, min_level=0, levels=2, eta=0.8, a_smooth=1.0, a_data=0.45)
    reg, flows = HipExecutor3D().process_batch(batch[:0], batch[:0], fixed, fixed, w0, None, None, flow_params=fp)
    assert reg.shape[0] == 0 and flows.shape == (0,) + batch.shape[1:4] + (3,)


def test_long_series_passes_through_the_device_in_windows(hip, monkeypatch):
    """fr3d_process_batch stages the series in windows of whole lock-step batches; a 200 KiB budget
    forces 9 volumes through as 4+4+1 and the result must equal the one-window run bit for bit."""
    from flowreg3d_amd.executor import HipExecutor3D
    fixed, batch = _series(T=9, shape=(10, 16, 20))
    w0 = np.zeros(batch.shape[1:4] + (3,), np.float32)
    fp = dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=10, min_level=0, levels=2, eta=0.8,
              a_smooth=1.0, a_data=0.45)
    calls = []
    with HipExecutor3D() as ex:
        reg1, fl1 = ex.process_batch(batch, batch, fixed, fixed, w0, None, None, flow_params=fp)
        monkeypatch.setenv("FR3D_STAGE_KIB", "200")
        reg2, fl2 = ex.process_batch(batch, batch, fixed, fixed, w0, None, None, flow_params=fp,
                                     progress_callback=calls.append)
    assert sum(calls) == 9
    assert np.array_equal(fl1, fl2) and np.array_equal(reg1, reg2)


def test_repeated_batches_are_deterministic_and_do_not_grow_device_memory(hip):
    """A long recording is many process_batch calls on one engine: same input -> same bits, and the
    workspace stops growing after the first call (all staging buffers are released again)."""
    import torch
    from flowreg3d_amd.executor import HipExecutor3D
    fixed, batch = _series(T=10, shape=(16, 24, 32))
    w0 = np.zeros(batch.shape[1:4] + (3,), np.float32)
    fp = dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=15, min_level=0, levels=3, eta=0.8,
              a_smooth=1.0, a_data=0.45)
    with HipExecutor3D() as ex:
        reg0, fl0 = ex.process_batch(batch, batch, fixed, fixed, w0, None, None, flow_params=fp)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        for _ in range(5):
            reg, fl = ex.process_batch(batch, batch, fixed, fixed, w0, None, None, flow_params=fp)
            assert np.array_equal(fl, fl0) and np.array_equal(reg, reg0)
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (8 << 20), (free0, free1)  # no growth beyond allocator granularity


def test_concurrent_callers_are_serialised_correctly(hip, oracle):
    """The reference's ThreadingExecutor3D calls get_displacement_func from several threads at once
    (parallelization/threading_3d.py:209-225); the engine serialises them on its mutex and every
    caller must get the result of its own input."""
    from concurrent.futures import ThreadPoolExecutor
    fixed, batch = _series(T=6, shape=(12, 18, 20))
    fp = dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=10, min_level=0, levels=2, eta=0.8,
              a_smooth=1.0, a_data=0.45)
    serial = [hip.get_displacement(fixed[..., 0], batch[t, ..., 0], **fp) for t in range(6)]
    with ThreadPoolExecutor(max_workers=4) as pool:
        threaded = list(pool.map(lambda t: hip.get_displacement(fixed[..., 0], batch[t, ..., 0], **fp), range(6)))
    for a, b in zip(serial, threaded):
        assert np.array_equal(a, b)
    assert not np.array_equal(serial[0], serial[5])


def test_errors_surface_as_exceptions_and_leave_the_engine_usable(hip):
    """A failing progress callback is re-raised after the batch, bad flow parameters are refused by the
    engine, and the executor keeps working afterwards (no lock or worker thread left behind)."""
    from flowreg3d_amd.executor import HipExecutor3D
    fixed, batch = _series(T=5, shape=(10, 14, 16))
    w0 = np.zeros(batch.shape[1:4] + (3,), np.float32)
    fp = dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=5, min_level=0, levels=2, eta=0.8,
              a_smooth=1.0, a_data=0.45)

    def boom(n):
        raise KeyError("callback failed")

    with HipExecutor3D() as ex:
        with pytest.raises(KeyError):
            ex.process_batch(batch, batch, fixed, fixed, w0, None, None, flow_params=fp, progress_callback=boom)
        with pytest.raises((ValueError, RuntimeError)):
            ex.process_batch(batch, batch, fixed, fixed, w0, None, None, flow_params=dict(fp, eta=1.5))
        with pytest.raises(NotImplementedError):
            ex.process_batch(batch, batch, fixed, fixed, w0, None, None, flow_params=dict(fp, cc_initialization=True))
        with pytest.raises(ValueError):
            ex.process_batch(batch, batch[:, :-1], fixed, fixed, w0, None, None, flow_params=fp)
        reg, flows = ex.process_batch(batch, batch, fixed, fixed, w0, None, None, flow_params=fp)
    assert np.isfinite(flows).all() and reg.shape == batch.shape


@pytest.mark.parametrize("C,a_smooth", [(1, 1.0), (2, 1.0), (1, 0.5)])
def test_two_engine_lanes_give_the_one_lane_results_bit_for_bit(hip, C, a_smooth):
    """Two engine lanes (the default; fr3d_set_lanes(1) = one): the lock-step batches of a series alternate between two
    engine lanes (two streams, two workspaces, lane 1 fed by a host thread of its own); 9 volumes at batch 4 run as
    2+2+2+2+1 on lanes 0,1,0,1,0.  Flows, registered volumes and the progress count must equal the one-lane run; a
    repeat on warm workspaces too."""
    from flowreg3d_amd import _lib
    from flowreg3d_amd.executor import HipExecutor3D
    fixed, batch = _series(T=9, shape=(12, 18, 22), C=C)
    w0 = np.zeros(batch.shape[1:4] + (3,), np.float32)
    fp = dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=12, min_level=0, levels=3, eta=0.8,
              a_smooth=a_smooth, a_data=0.45)
    lib = _lib.init(0)
    calls = []
    with HipExecutor3D() as ex:
        lib.fr3d_set_batch(4)
        try:
            assert lib.fr3d_set_lanes(1) == 2  # two lanes are the default
            reg1, fl1 = ex.process_batch(batch, batch, fixed, fixed, w0, None, None, flow_params=fp)
            assert lib.fr3d_set_lanes(2) == 1
            for rep in range(2):
                calls.clear()
                reg2, fl2 = ex.process_batch(batch, batch, fixed, fixed, w0, None, None, flow_params=fp,
                                             progress_callback=calls.append)
                assert sum(calls) == 9
                assert np.array_equal(fl1, fl2) and np.array_equal(reg1, reg2), rep
        finally:
            lib.fr3d_set_lanes(2)
            lib.fr3d_set_batch(0)


def test_progress_callback_runs_on_the_callers_thread_and_may_reenter_the_library(hip):
    """Two engine lanes: lane 1 is fed by a thread of the library, but the progress callback is delivered on the CALLER's
    thread (where the reference's executors call it, and where a GUI progress bar lives) -- and it may call back into the
    library (the caller's thread holds the library's recursive lock; from another thread that would deadlock)."""
    import threading
    from flowreg3d_amd import _lib
    from flowreg3d_amd.executor import HipExecutor3D
    fixed, batch = _series(T=6)
    w0 = np.zeros(batch.shape[1:4] + (3,), np.float32)
    fp = dict(alpha=(0.25,) * 3, update_lag=5, iterations=5, min_level=0, levels=2, eta=0.8, a_smooth=1.0, a_data=0.45)
    lib = _lib.load()
    prev = lib.fr3d_set_lanes(2)
    seen = []

    def cb(k):
        seen.append((int(k), threading.get_ident()))
        assert lib.fr3d_sync() == 0          # re-enters the library
        lib.fr3d_last_error()

    try:
        lib.fr3d_set_batch(2)                # 6 volumes as chunks of 1 on two lanes
        HipExecutor3D().process_batch(batch, batch, fixed, fixed, w0, None, None, progress_callback=cb, flow_params=fp)
    finally:
        lib.fr3d_set_batch(0)
        lib.fr3d_set_lanes(prev)
    assert sum(k for k, _ in seen) == 6
    assert {t for _, t in seen} == {threading.get_ident()}


def test_tight_memory_budget_odd_batch_one_lane_then_two_lanes(hip):
    """the two-lane split must budget for the layout that runs.  A device that "has room" for three volumes
    (FR3D_MEM_CAP_MIB) runs them as 1 + 1 + 1 on two lanes, not 2 + 2; a one-lane call (profiling brackets on) before it
    leaves slabs sized for three volumes behind, which are given back; results equal the uncapped run's."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from flowreg3d_amd import _lib
from flowreg3d_amd.executor import HipExecutor3D
from flowreg3d_amd.synthetic import make_pair
shape = (40, 48, 44)
fixed, _, _ = make_pair(shape, seed=3)
batch = np.stack([make_pair(shape, seed=3, scale=s)[1] for s in (0.2, 0.4, 0.6, 0.8, 1.0)])[..., None]
ref = fixed[..., None]
w0 = np.zeros(shape + (3,), np.float32)
fp = dict(alpha=(0.25,) * 3, update_lag=5, iterations=10, min_level=0, levels=2, eta=0.8, a_smooth=1.0, a_data=0.45, solver_fp64=2)
lib = _lib.init(0)
ex = HipExecutor3D()
lib.fr3d_prof_enable(1)                       # one lane: slabs for the whole budgeted batch
_, f1 = ex.process_batch(batch, batch, ref, ref, w0, None, None, flow_params=fp)
lib.fr3d_prof_enable(0)                       # two lanes
_, f2 = ex.process_batch(batch, batch, ref, ref, w0, None, None, flow_params=fp)
np.save(sys.argv[1], np.stack([f1, f2]))
'''
    import tempfile
    outs = {}
    with tempfile.TemporaryDirectory() as td:
        for tag in ("free", "tight"):
            env = dict(os.environ)
            if tag == "tight":
                env["FR3D_MEM_CAP_MIB"] = str(outs["cap"])  # room for three volumes' solver slabs
            path = os.path.join(td, tag + ".npy")
            r = subprocess.run([sys.executable, "-c", code, path], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            outs[tag] = np.load(path)
            if tag == "free":
                nfin = 40 * 48 * 44
                per_vol = 1.3 * nfin * 8.0 * (12 + 9 + 6) + nfin * 36.0   # solver_bytes_per_volume's formula, roughly
                outs["cap"] = int(3.4 * per_vol / 1048576.0) + 1
    assert np.array_equal(outs["free"][0], outs["free"][1])          # one lane == two lanes
    assert np.array_equal(outs["tight"][0], outs["free"][0])
    assert np.array_equal(outs["tight"][1], outs["free"][0])


def test_small_volumes_run_in_large_lockstep_batches_with_the_single_call_results(hip):
    """Round 4: the default lock-step batch holds the same number of VOXELS for small volumes as 8 volumes of 256^3 (up
    to 128 volumes; engine.hip: batch_wanted).  41 small volumes in one call -- two lanes of 20 / 21, far above the old
    8 -- must give the flows of 41 single get_displacement calls bit for bit (blockIdx.y = volume of the batch)."""
    from flowreg3d_amd.executor import HipExecutor3D
    from flowreg3d_amd.synthetic import make_pair
    shape = (10, 14, 18)
    fixed, _, _ = make_pair(shape, seed=3)
    T = 41
    batch = np.stack([make_pair(shape, seed=3, scale=0.02 * (t + 1))[1] for t in range(T)])[..., None].astype(np.float32)
    ref = fixed[..., None].astype(np.float32)
    fp = dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=12, min_level=0, levels=3, eta=0.8, a_smooth=1.0,
              a_data=0.45)
    calls = []
    _, flows = HipExecutor3D().process_batch(batch, batch.astype(np.float64), ref, ref.astype(np.float64),
                                             np.zeros(shape + (3,), np.float32), None, None,
                                             progress_callback=lambda k: calls.append(int(k)), flow_params=fp)
    assert flows.shape == (T,) + shape + (3,) and sum(calls) == T
    for t in (0, 7, 8, 20, 21, 40):
        one = hip.get_displacement(ref, batch[t], **fp)
        assert np.array_equal(flows[t], one.astype(np.float32)), t
