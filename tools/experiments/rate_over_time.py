#!/usr/bin/env python3
"""SOR rate of consecutive cfg2 batches over time in ONE process (no re-allocation), with rocm-smi clocks, power
and temperatures sampled beside it: is the 0.48 / 0.51 spread of the 256^3 line a device state that comes and goes?
usage (GPU box): python tools/experiments/rate_over_time.py [seconds]"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowreg3d_amd import _lib  # noqa: E402
from flowreg3d_amd.synthetic import fast_pair  # noqa: E402

dur = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
n, nb = 256, 8
lib = _lib.init(0)
fixed, moving, _ = fast_pair((n, n, n))
nv = n ** 3
params = _lib.make_params(alpha=(0.25,) * 3, update_lag=5, iterations=100, min_level=0, levels=4, eta=0.8,
                          a_smooth=1.0, a_data=0.45, n_channels=1, solver_fp64=None)
ref = lib.fr3d_dev_malloc(nv * 4); mov = lib.fr3d_dev_malloc(nv * 4 * nb)
flows = lib.fr3d_dev_malloc(nv * 12 * nb); regs = lib.fr3d_dev_malloc(nv * 4 * nb)
lib.fr3d_h2d(ref, fixed.ctypes.data, nv * 4)
for b in range(nb):
    lib.fr3d_h2d(mov + b * nv * 4, moving.ctypes.data, nv * 4)
lib.fr3d_set_batch(nb)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
        keep = [" ".join(l.split()) for l in out.splitlines() if any(k in l for k in ("sclk", "fclk", "mclk", "Power", "junction", "memory)"))]
        return "; ".join(keep)
    except Exception as e:  # noqa: BLE001
        return f"smi failed: {e}"


t_start = time.time()
last_smi = -100.0
while time.time() - t_start < dur:
    lib.fr3d_prof_enable(1); lib.fr3d_prof_reset()
    _lib.check(lib.fr3d_process_batch_dev(C.byref(params), mov, mov, ref, ref, None, None, nb, n, n, n, 1, 3, flows, regs,
                                          C.cast(None, _lib.PROGRESS_FN), None))
    lib.fr3d_sync()
    s = _lib.prof_get()["sor"]
    now = time.time() - t_start
    rec = {"t": round(now, 1), "frac": round(s["algo_bytes"] / s["ms"] / 8e9, 4), "sor_ms_per_vol": round(s["ms"] / nb, 2)}
    if now - last_smi > 10:
        rec["smi"] = smi()
        last_smi = now
    print(json.dumps(rec), flush=True)
