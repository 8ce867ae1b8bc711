#!/usr/bin/env python3
"""the verification sweep alone (fr3d_level_solve_verify) against the ppow oracle's compute_flow_3d on one level"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowreg3d_amd import _lib
from flowreg3d_amd.synthetic import make_pair
from oracle import oracle
oracle.build(); oracle.use_build("ppow")
lib = _lib.init(0)
shape = tuple(int(v) for v in sys.argv[1].split(","))
its = int(sys.argv[2]); lag = int(sys.argv[3]) if len(sys.argv) > 3 else 5
fixed, moving, gt = make_pair(shape, seed=7, cheap=True)
Z, Y, X = shape
J = oracle.get_motion_tensor_gc(fixed, moving, 1.0, 1.0, 1.0)          # 10 x (Z+2,Y+2,X+2)
rng = np.random.default_rng(1)
uvw = (0.8 * np.moveaxis(gt, -1, 0) + 0.02 * rng.standard_normal((3, Z, Y, X))).astype(np.float32)
pad = lambda a: np.pad(a.astype(np.float64), 1, mode="edge")
wt = np.zeros((Z + 2, Y + 2, X + 2, 1)); wt[1:-1, 1:-1, 1:-1, 0] = 1.0
want = oracle.compute_flow_3d(*[j[..., None] for j in J], wt, pad(uvw[0]), pad(uvw[1]), pad(uvw[2]), 0.25, 0.25, 0.25,
                              its, lag, 0.45, 1.0, 1.0, 1.0, 1.0)[1:-1, 1:-1, 1:-1]
Jin = np.ascontiguousarray(np.stack([j[1:-1, 1:-1, 1:-1] for j in J])[None])   # (1,10,Z,Y,X)
w32 = np.ones((1, Z, Y, X), np.float32)
out = np.empty((3, Z, Y, X), np.float64)
al = (C.c_double * 3)(0.25, 0.25, 0.25); ad = (C.c_double * 1)(0.45)
_lib.check(lib.fr3d_level_solve_verify(_lib.ptr(Jin), _lib.ptr(w32), _lib.ptr(uvw), Z, Y, X, 1, al, its, lag, ad, 1.0, 1.0, 1.0, _lib.ptr(out)))
got = np.moveaxis(out, 0, -1)
d = np.abs(got - want)
idx = np.argwhere(d.max(axis=-1) > 0)
print(shape, "its", its, "lag", lag, "differ", int((d > 0).sum()), "of", d.size, "max", float(d.max()),
      "first voxels", idx[:5].tolist(), flush=True)
