"""Does the 512^3 SOR rate depend on the allocation (re-init inside one process) or on the process?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flowreg3d_amd as fr
from flowreg3d_amd import _lib
from flowreg3d_amd.synthetic import fast_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
fixed, moving, _ = fast_pair((n, n, n))
kw = dict(alpha=(0.25,) * 3, update_lag=5, iterations=100, min_level=0, levels=5 if n >= 512 else 4, eta=0.8,
          a_smooth=1.0, a_data=0.45)
lib = _lib.init(0)
for rep in range(reps):
    for inner in range(2):
        lib.fr3d_prof_enable(1); lib.fr3d_prof_reset()
        fr.get_displacement(fixed, moving, **kw)
        st = _lib.prof_get()["sor"]
        print("alloc %d run %d: sor %.1f ms  %.0f GB/s" % (rep, inner, st["ms"], st["algo_bytes"] / st["ms"] / 1e6), flush=True)
        lib.fr3d_prof_enable(0)
    _lib.shutdown()
    lib = _lib.init(0)
