"""Timing of the a_smooth != 1 solver path against the fast path (128^3, 4-level pyramid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flowreg3d_amd as fr
from flowreg3d_amd import _lib
from flowreg3d_amd.synthetic import fast_pair
lib = _lib.init(0)
f, m, g = fast_pair((128, 128, 128))
for asm in (1.0, 0.5):
    kw = dict(alpha=(0.25,) * 3, update_lag=5, iterations=100, min_level=0, levels=3, eta=0.8, a_smooth=asm, a_data=0.45)
    fr.get_displacement(f, m, **kw)
    lib.fr3d_prof_enable(1); lib.fr3d_prof_reset()
    t = time.time(); fr.get_displacement(f, m, **kw); dt = time.time() - t
    st = _lib.prof_get()["sor"]
    print("a_smooth %.1f: total %.1f ms, sor %.1f ms, launches %d" % (asm, dt * 1e3, st["ms"], st["launches"]), flush=True)
    lib.fr3d_prof_enable(0)
