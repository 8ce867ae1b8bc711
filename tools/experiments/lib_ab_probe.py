#!/usr/bin/env python3
"""A/B of two builds of the engine (FR3D_LIB selects the shared library; one process per build, alternated):
SOR time per volume and the whole step; LIB may carry one environment setting as path.so@VAR=value.
usage (GPU box): [FR3D_PROBE_MODE=0..3] python tools/experiments/lib_ab_probe.py EDGE BATCH REPS LIB_A LIB_B [LIB_C ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, %r)
from flowreg3d_amd import _lib
from flowreg3d_amd.synthetic import fast_pair
n, nb = int(sys.argv[1]), int(sys.argv[2])
lib = _lib.init(0)
levels = {64: 2, 128: 3, 256: 4, 512: 5}[n]
fixed, moving, _ = fast_pair((n, n, n))
nc = int(os.environ.get("FR3D_PROBE_CHANNELS", "1"))
if nc > 1:
    import numpy as np
    fixed = np.ascontiguousarray(np.stack([fixed] + [fixed[::-1] * 0.9 for _ in range(nc - 1)], -1))
    moving = np.ascontiguousarray(np.stack([moving] + [moving[::-1] * 0.9 for _ in range(nc - 1)], -1))
nv = n ** 3 * nc
params = _lib.make_params(alpha=(0.25,) * 3, update_lag=5, iterations=100, min_level=0, levels=levels, eta=0.8,
                          a_smooth=1.0, a_data=0.45, n_channels=nc,
                          solver_fp64=int(os.environ.get("FR3D_PROBE_MODE", "1")))
ref = lib.fr3d_dev_malloc(nv * 4); mov = lib.fr3d_dev_malloc(nv * 4 * nb)
flows = lib.fr3d_dev_malloc(n ** 3 * 12 * nb); regs = lib.fr3d_dev_malloc(nv * 4 * nb)
lib.fr3d_h2d(ref, fixed.ctypes.data, nv * 4)
for b in range(nb):
    lib.fr3d_h2d(mov + b * nv * 4, moving.ctypes.data, nv * 4)
lib.fr3d_set_batch(nb)
def run(prof):
    lib.fr3d_prof_enable(1 if prof else 0)
    if prof: lib.fr3d_prof_reset()
    _lib.check(lib.fr3d_process_batch_dev(C.byref(params), mov, mov, ref, ref, None, None, nb, n, n, n, nc, 3, flows, regs,
                                          C.cast(None, _lib.PROGRESS_FN), None))
    lib.fr3d_sync()
t0 = time.time()
while time.time() - t0 < 10: run(False)
best = None
for _ in range(3):
    t0 = time.perf_counter(); run(True); wall = time.perf_counter() - t0
    s = _lib.prof_get()
    r = {"sor_ms_per_vol": round(s["sor"]["ms"] / nb, 2), "frac": round(s["sor"]["algo_bytes"] / s["sor"]["ms"] / 8e9, 4),
         "tensor_ms": round(s["tensor"]["ms"] / nb, 2), "other_ms": round(s["other"]["ms"] / nb, 2), "wall_ms_per_vol": round(1e3 * wall / nb, 1)}
    if best is None or r["wall_ms_per_vol"] < best["wall_ms_per_vol"]: best = r
import hashlib, numpy as np
out = np.empty((n, n, n, 3), np.float32); lib.fr3d_d2h(out.ctypes.data, flows, n ** 3 * 12)
best["flow_sha"] = hashlib.sha256(out.tobytes()).hexdigest()[:12]
st = C.c_double(0.0)
lib.fr3d_stream_probe(1 << 28, 10, C.byref(st))
best["stream_GBs"] = round(st.value)  # ~5300: the box state in which the 256^3 sweep runs at 0.51; ~5700: 0.48
print(json.dumps(best))
""" % ROOT


def main():
    n, nb, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    specs = sys.argv[4:]
    for rep in range(reps):
        for tag, spec in zip("ABCDEFGH", specs):
            lib, _, setting = spec.partition("@")  # "path/to/lib.so@VAR=value" sets VAR for that side
            env = dict(os.environ, FR3D_LIB=os.path.abspath(lib))
            if setting:
                k, _, v = setting.partition("=")
                env[k] = v
            r = subprocess.run([sys.executable, "-c", CHILD, n, nb], env=env, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            print(json.dumps({"edge": int(n), "batch": int(nb), "mode": int(os.environ.get("FR3D_PROBE_MODE", "1")), "lib": tag + ":" + os.path.basename(spec), "rep": rep,
                              **(json.loads(line[-1]) if line else {"error": r.stderr[-300:]})}), flush=True)


if __name__ == "__main__":
    main()
