#!/usr/bin/env python3
"""verification mode vs the ppow oracle on growing sizes (where does bit-identity stop, if it does?)"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import flowreg3d_amd as fr
from flowreg3d_amd.synthetic import make_pair
from oracle import oracle
oracle.build(); oracle.use_build("ppow")
cases = [((48, 64, 80), 1, 3), ((64, 96, 128), 1, 4), ((64, 96, 128), 2, 4), ((96, 128, 160), 1, 4), ((40, 300, 60), 1, 3), ((128, 160, 192), 1, 5)]
if len(sys.argv) > 2:  # "Z,Y,X,C,levels;..."
    cases = [((int(a), int(b), int(c)), int(d), int(e)) for a, b, c, d, e in (q.split(",") for q in sys.argv[2].split(";"))]
for shape, ch, levels in cases:
    fixed, moving, _ = make_pair(shape, seed=7, channels=ch, cheap=True)
    kw = dict(alpha=(0.25,) * 3, update_lag=5, iterations=int(sys.argv[1]) if len(sys.argv) > 1 else 15,
              min_level=int(os.environ.get("VS_MIN_LEVEL", "0")), levels=levels, eta=0.8, a_smooth=1.0,
              a_data=float(os.environ.get("VS_A_DATA", "0.45")))
    t0 = time.time(); want = oracle.get_displacement(fixed, moving, **kw); t1 = time.time()
    got = fr.get_displacement_verify(fixed, moving, **kw)
    d = np.abs(got - want)
    print(shape, ch, levels, "differ", int((d > 0).sum()), "of", d.size, "max", float(d.max()), f"oracle {t1-t0:.0f}s", flush=True)
