#!/usr/bin/env python3
"""Volumes beyond the BASELINE sizes: does the path hold where element offsets pass 2^31 (a 768^3 slab of 12-value records
is 5.4e9 storage elements) and where one volume's slabs are most of the device?  A translated stand-in texture
(synthetic.fast_pair), default solver mode, a short pyramid; checks that the flow is finite and recovers the translation.
   tools/experiments/big_volume_probe.py Z Y X [iterations] [levels]
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowreg3d_amd import core, _lib
from flowreg3d_amd.synthetic import fast_pair, epe

Z, Y, X = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
levels = int(sys.argv[5]) if len(sys.argv) > 5 else 6
t0 = time.time()
fixed, moving, gt = fast_pair((Z, Y, X))
t_gen = time.time() - t0
lib = _lib.init(0)
kw = dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=iters, levels=levels, min_level=0, eta=0.5, a_smooth=1.0,
          a_data=0.45)
t0 = time.time()
flow = core.get_displacement(fixed, moving, **kw)
t_run = time.time() - t0
m, mx = epe(flow, gt, crop=16)
print(json.dumps({"shape": [Z, Y, X], "voxels": Z * Y * X, "generate_s": round(t_gen, 1), "call_s": round(t_run, 2),
                  "solver_mode": lib.fr3d_last_solver_mode(), "fallback": lib.fr3d_last_solver_fallback(),
                  "finite": bool(np.isfinite(flow).all()), "epe_vs_translation_mean_crop16": m, "max": mx,
                  "flow_mean": [float(flow[..., c].mean()) for c in range(3)]}), flush=True)
assert np.isfinite(flow).all() and m < 0.2, "flow does not recover the translation"
