"""Randomised HipExecutor3D.process_batch calls (shapes, T = 0..6, one or two channels, float32 /
float64 / uint16 batches, cubic / linear) against the sequential executor body restated on the oracle.
usage (GPU box): python tools/fuzz_executor.py [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowreg3d_amd.executor import HipExecutor3D
from oracle import oracle
from scipy.ndimage import gaussian_filter
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
with HipExecutor3D() as ex:
    for case in range(40):
        shape = tuple(int(v) for v in rng.choice([2, 5, 7, 9, 16, 21, 33], size=3))
        C = int(rng.choice([1, 2])); T = int(rng.integers(0, 7))
        dt = rng.choice([np.float32, np.float64, np.uint16])
        method = str(rng.choice(["cubic", "linear"]))
        def vol():
            a = gaussian_filter(rng.random(shape + (C,)), (1, 1, 1, 0), mode="reflect")
            return (a - a.min()) / (a.max() - a.min() + 1e-12)
        ref = vol()
        proc = np.stack([0.97 * gaussian_filter(ref, (0.5, 0.5, 0.5, 0)) + 0.02 * rng.random() for _ in range(T)]) if T else np.zeros((0,) + shape + (C,))
        raw = (proc * 1000).astype(dt) if dt == np.uint16 else proc.astype(dt)
        ref_raw = ref * 1000 if dt == np.uint16 else ref
        w0 = (0.2 * rng.standard_normal(shape + (3,))).astype(np.float32)
        fp = dict(alpha=(0.5, 0.5, 0.5), update_lag=int(rng.integers(1, 6)), iterations=int(rng.integers(1, 15)), min_level=0,
                  levels=int(rng.integers(1, 5)), eta=0.8, a_smooth=1.0, a_data=0.45)
        calls = []
        reg, flows = ex.process_batch(raw, proc, ref_raw, ref, w0, None, None, interpolation_method=method,
                                      progress_callback=calls.append, flow_params=fp)
        ok = reg.shape == raw.shape and reg.dtype == raw.dtype and flows.shape == (T,) + shape + (3,) and sum(calls) == T
        for t in range(T):
            f = oracle.get_displacement(ref, proc[t], uvw=w0.copy(), **fp).astype(np.float32)
            r = oracle.imregister_wrapper(raw[t], f[..., 0], f[..., 1], f[..., 2], ref_raw, method)
            want = np.empty_like(raw[t]); want[...] = np.asarray(r).reshape(want.shape)
            e = np.linalg.norm(flows[t].astype(np.float64) - f, axis=-1).mean()
            dr = np.abs(reg[t].astype(np.float64) - want.astype(np.float64)).max()
            tol = 1.0 if dt == np.uint16 else 5e-4
            ok = ok and e < 1e-4 and dr <= tol
        bad += not ok
        print("ok " if ok else "BAD", case, shape, "C", C, "T", T, np.dtype(dt).name, method, flush=True)
print("bad", bad)
