"""Timing of the multi-channel solver modes (half-scale BASELINE config 5)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flowreg3d_amd as fr
from flowreg3d_amd import _lib
from flowreg3d_amd.synthetic import fast_pair
lib = _lib.init(0)
f, m, _ = fast_pair((128, 256, 256))
f2, m2, _ = fast_pair((128, 256, 256), seed=77)
fixed = np.stack([f, f2], -1); moving = np.stack([m, m2], -1)
kw = dict(alpha=(0.25,) * 3, update_lag=5, iterations=100, min_level=0, levels=8, eta=0.8, a_smooth=1.0, a_data=0.45,
          weight=np.array([0.5, 0.5]))
for mode in (0, 2):
    fr.get_displacement(fixed, moving, solver_fp64=mode, **kw)
    lib.fr3d_prof_enable(1); lib.fr3d_prof_reset()
    t = time.time(); fr.get_displacement(fixed, moving, solver_fp64=mode, **kw); dt = time.time() - t
    st = _lib.prof_get()["sor"]
    print("C=2 solver_fp64=%d: total %.1f ms, sor %.1f ms (%.0f GB/s algorithmic at %d B/update)" %
          (mode, dt * 1e3, st["ms"], st["algo_bytes"] / st["ms"] / 1e6, 4 * 29), flush=True)
    lib.fr3d_prof_enable(0)
