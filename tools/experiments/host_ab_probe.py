#!/usr/bin/env python3
"""A/B of two builds on the HOST-array entry (NumPy in, NumPy out through HipExecutor3D.process_batch, 8 volumes of
256^3): one process per build, alternated; each warms the device for ~12 s, then reports the best of 4 calls.
usage (GPU box): python tools/experiments/host_ab_probe.py LIB_A LIB_B [reps]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import json, os, sys, time
sys.path.insert(0, %r)
import numpy as np
import bench
from flowreg3d_amd.executor import HipExecutor3D
from flowreg3d_amd.synthetic import fast_pair
Z = Y = X = 256
fixed, moving, _ = fast_pair((Z, Y, X))
batch = np.ascontiguousarray(np.stack([moving] * 8)[..., None])
fp = dict(bench.solver_kwargs(4), weight=np.array([1.0]), solver_fp64=None)
w0 = np.zeros((Z, Y, X, 3), np.float32)
ref = fixed[..., None]
with HipExecutor3D() as ex:
    t0 = time.time()
    while time.time() - t0 < 12:
        ex.process_batch(batch, batch, ref, ref, w0, None, None, flow_params=fp)
    best = None
    for _ in range(4):
        t0 = time.perf_counter()
        reg, flows = ex.process_batch(batch, batch, ref, ref, w0, None, None, flow_params=fp)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        del reg, flows
print(json.dumps({"volumes_per_s": round(8 / best, 3), "s_per_batch": round(best, 4)}))
""" % ROOT

for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
    for tag, lib in (("A", sys.argv[1]), ("B", sys.argv[2])):
        env = dict(os.environ, FR3D_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(tag, os.path.basename(lib), line[-1] if line else r.stderr[-500:], flush=True)
