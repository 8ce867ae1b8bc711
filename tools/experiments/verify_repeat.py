#!/usr/bin/env python3
"""is the verification mode deterministic on the device, and independent of the tile shape?"""
import os, sys, hashlib, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
import flowreg3d_amd as fr
from flowreg3d_amd.synthetic import make_pair
shape = tuple(int(v) for v in sys.argv[1].split(","))
fixed, moving, _ = make_pair(shape, seed=7, cheap=True)
kw = dict(alpha=(0.25,) * 3, update_lag=5, iterations=int(sys.argv[2]), min_level=0, levels=int(sys.argv[3]), eta=0.8, a_smooth=1.0, a_data=0.45)
for rep in range(2):
    f = fr.get_displacement_verify(fixed, moving, **kw)
    print("HASH", hashlib.sha256(f.tobytes()).hexdigest()[:16])
""" % ROOT
for shp in (None, "1x1", "4x1"):
    env = dict(os.environ)
    if shp:
        env["FR3D_SOR_SHAPE"] = shp
        env["FR3D_LIB"] = os.path.join(ROOT, "flowreg3d_amd", "lib", "libflowreg3d_hip_exp.so")
    r = subprocess.run([sys.executable, "-c", CHILD] + sys.argv[1:4], env=env, capture_output=True, text=True)
    print(shp, [l for l in r.stdout.splitlines() if l.startswith("HASH")], r.stderr[-200:] if r.returncode else "")
