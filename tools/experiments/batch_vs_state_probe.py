#!/usr/bin/env python3
"""SOR rate at 256^3 per lock-step batch size, with the y += x stream rate of the same process (which tells the
box's state: ~5.3 TB/s = the state in which the sweep runs at 0.51, ~5.7 = the one in which it runs at 0.48).
usage (GPU box): python tools/experiments/batch_vs_state_probe.py"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowreg3d_amd import _lib  # noqa: E402
from flowreg3d_amd.synthetic import fast_pair  # noqa: E402

n, NB = 256, 16
lib = _lib.init(0)
fixed, moving, _ = fast_pair((n, n, n))
nv = n ** 3
params = _lib.make_params(alpha=(0.25,) * 3, update_lag=5, iterations=100, min_level=0, levels=4, eta=0.8,
                          a_smooth=1.0, a_data=0.45, n_channels=1, solver_fp64=None)
ref = lib.fr3d_dev_malloc(nv * 4); mov = lib.fr3d_dev_malloc(nv * 4 * NB)
flows = lib.fr3d_dev_malloc(nv * 12 * NB); regs = lib.fr3d_dev_malloc(nv * 4 * NB)
lib.fr3d_h2d(ref, fixed.ctypes.data, nv * 4)
for b in range(NB):
    lib.fr3d_h2d(mov + b * nv * 4, moving.ctypes.data, nv * 4)


def run(nb, prof):
    lib.fr3d_set_batch(nb)
    lib.fr3d_prof_enable(1 if prof else 0)
    if prof:
        lib.fr3d_prof_reset()
    _lib.check(lib.fr3d_process_batch_dev(C.byref(params), mov, mov, ref, ref, None, None, NB, n, n, n, 1, 3, flows, regs,
                                          C.cast(None, _lib.PROGRESS_FN), None))
    lib.fr3d_sync()


t0 = time.time()
while time.time() - t0 < 12:
    run(8, False)
for rep in range(2):
    for nb in (16, 8, 4, 2):
        run(nb, False)
        t0 = time.perf_counter(); run(nb, True); wall = time.perf_counter() - t0
        s = _lib.prof_get()["sor"]
        print(json.dumps({"batch": nb, "frac": round(s["algo_bytes"] / s["ms"] / 8e9, 4), "sor_ms_per_vol": round(s["ms"] / NB, 2),
                          "wall_ms_per_vol": round(1e3 * wall / NB, 2)}), flush=True)
st = C.c_double(0.0)
_lib.check(lib.fr3d_stream_probe(1 << 28, 20, C.byref(st)))
print(json.dumps({"stream_y_plus_x_GBs": round(st.value, 1)}))
