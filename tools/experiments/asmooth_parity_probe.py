#!/usr/bin/env python3
"""Full-size parity of the psi_smooth solver path (a_smooth = 0.5 on config 2's geometry) against the committed
CPU-oracle sample tests/golden/fullsize_cfg2_asmooth05.npz, per solver mode.  GPU box: python tools/experiments/asmooth_parity_probe.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import flowreg3d_amd as fr  # noqa: E402
from flowreg3d_amd.synthetic import fullsize_case  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_cfg2_asmooth05.npz"))
meta = json.loads(bytes(g["meta"]).decode())
fixed, moving, gt, kw = fullsize_case("cfg2_asmooth05")
st, bl = meta["stride"], meta["block"]
z0, y0, x0 = meta["block_origin_zyx"]


def epe(a, b):
    d = np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64), axis=-1)
    return float(d.mean()), float(d.max())


for mode in (None, 2, 1):
    t0 = time.time()
    flow = fr.get_displacement(fixed, moving, solver_fp64=mode, **kw)
    dt = time.time() - t0
    lm, lx = epe(flow[::st, ::st, ::st], g["lattice"])
    bm, bx = epe(flow[z0:z0 + bl, y0:y0 + bl, x0:x0 + bl], g["block"])
    print(json.dumps({"mode": mode, "lattice_mean": lm, "lattice_max": lx, "block_mean": bm, "block_max": bx, "seconds": round(dt, 2)}), flush=True)
