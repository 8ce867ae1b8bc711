#!/usr/bin/env python3
"""Timeline of the host-array entry (HipExecutor3D.process_batch on NumPy arrays, 8 volumes of 256^3): Python-side
preparation, the C call (experiment build prints its phases with FR3D_HOST_TRACE=1), the device-resident rate beside it.
usage (GPU box): FR3D_LIB=flowreg3d_amd/lib/libflowreg3d_hip_exp.so FR3D_HOST_TRACE=1 python tools/experiments/host_path_trace.py [T]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from flowreg3d_amd.executor import HipExecutor3D  # noqa: E402
from flowreg3d_amd.synthetic import fast_pair  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Z, Y, X, levels, _ = bench.WORKLOADS["cfg2"]
fixed, moving, _ = fast_pair((Z, Y, X))
batch = np.ascontiguousarray(np.stack([moving] * T)[..., None])
fp = dict(bench.solver_kwargs(levels), weight=np.array([1.0]))
w0 = np.zeros((Z, Y, X, 3), np.float32)
ref = fixed[..., None]
from flowreg3d_amd import _lib  # noqa: E402

lib = _lib.init(0)
c_call = lib.fr3d_process_batch_raw
spent = {}


def timed_c_call(*a):
    t = time.perf_counter()
    rc = c_call(*a)
    spent["c"] = time.perf_counter() - t
    return rc


class _Proxy:  # the executor's library handle with the batch entry timed
    def __getattr__(self, name):
        return timed_c_call if name == "fr3d_process_batch_raw" else getattr(lib, name)


with HipExecutor3D() as ex:
    ex._lib = _Proxy()
    reg = flows = None
    for rep in range(3):
        print(f"--- call {rep}", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        del reg, flows
        t_free = time.perf_counter() - t0
        t0 = time.perf_counter()
        out = ex.process_batch(batch, batch, ref, ref, w0, None, None, flow_params=fp)
        dt = time.perf_counter() - t0
        reg, flows = out
        del out
        print(f"call {rep}: {1e3 * dt:.1f} ms for {T} volumes = {T / dt:.2f} volumes/s; inside the C entry {1e3 * spent['c']:.1f} ms, "
              f"Python around it {1e3 * (dt - spent['c']):.1f} ms; releasing the previous results took {1e3 * t_free:.1f} ms", flush=True)
