#!/usr/bin/env python3
"""level 0 of the failing case, stage by stage, on the actual data"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import flowreg3d_amd as fr
from flowreg3d_amd import _lib
from flowreg3d_amd.synthetic import make_pair
from oracle import oracle
oracle.build(); oracle.use_build("ppow")
lib = _lib.init(0)
shape = (128, 160, 192)
fixed, moving, _ = make_pair(shape, seed=7, cheap=True)
kw = dict(alpha=(0.25,) * 3, update_lag=5, iterations=15, min_level=1, levels=5, eta=0.8, a_smooth=1.0, a_data=0.45)
u_o = oracle.get_displacement(fixed, moving, **kw)          # = resize(u_level1) : level-0 initial flow (fp32 values)
u_g = fr.get_displacement_verify(fixed, moving, **kw)
print("u_init identical", np.array_equal(u_o, u_g), "fp32-exact", np.array_equal(u_o, u_o.astype(np.float32)))
u = u_o.astype(np.float32)
w_o = oracle.imregister_wrapper(moving, u[..., 0], u[..., 1], u[..., 2], fixed)
w_g = fr.imregister_wrapper(moving, u[..., 0], u[..., 1], u[..., 2], fixed)
d = w_o.astype(np.float64) - w_g.astype(np.float64)
print("warp: differ", int((d != 0).sum()), "of", d.size, "max", float(np.abs(d).max()), "first", np.argwhere(d != 0)[:3].tolist())
# tensor on the oracle's warp
J = oracle.get_motion_tensor_gc(fixed, w_o, 1.0, 1.0, 1.0)
Z, Y, X = shape
got = np.empty((10, Z, Y, X), np.float64)
_lib.check(lib.fr3d_motion_tensor_f64(_lib.ptr(np.ascontiguousarray(fixed, np.float32)), _lib.ptr(np.ascontiguousarray(w_o, np.float32)), Z, Y, X, 1.0, 1.0, 1.0, _lib.ptr(got)))
want = np.stack([j[1:-1, 1:-1, 1:-1] for j in J])
print("tensor: differ", int((got != want).sum()))
# sweep on the oracle's tensor and the real u
pad = lambda a: np.pad(a.astype(np.float64), 1, mode="edge")
wt = np.zeros((Z + 2, Y + 2, X + 2, 1)); wt[1:-1, 1:-1, 1:-1, 0] = 1.0
ref = oracle.compute_flow_3d(*[j[..., None] for j in J], wt, pad(u[..., 0]), pad(u[..., 1]), pad(u[..., 2]), 0.25, 0.25, 0.25,
                             15, 5, 0.45, 1.0, 1.0, 1.0, 1.0)[1:-1, 1:-1, 1:-1]
out = np.empty((3, Z, Y, X), np.float64)
al = (C.c_double * 3)(0.25, 0.25, 0.25); ad = (C.c_double * 1)(0.45)
uvw = np.ascontiguousarray(np.moveaxis(u, -1, 0))
Jin = np.ascontiguousarray(want[None]); ones = np.ones((1, Z, Y, X), np.float32)
_lib.check(lib.fr3d_level_solve_verify(_lib.ptr(Jin), _lib.ptr(ones), _lib.ptr(uvw), Z, Y, X, 1, al, 15, 5, ad, 1.0, 1.0, 1.0, _lib.ptr(out)))
d = np.abs(np.moveaxis(out, 0, -1) - ref)
print("sweep: differ", int((d > 0).sum()), "max", float(d.max()), "first", np.argwhere(d.max(axis=-1) > 0)[:3].tolist())

# median + accumulate in fp64 on the oracle's increments: oracle's median vs the final flows of both sides
med = np.stack([oracle.median5(ref[..., d]) for d in range(3)], -1)
flow_o = u.astype(np.float64) + med
kw0 = dict(kw, min_level=0)
full_o = oracle.get_displacement(fixed, moving, **kw0)
full_g = fr.get_displacement_verify(fixed, moving, **kw0)
print("oracle stages reassembled == oracle full:", np.array_equal(flow_o, full_o), " == gpu full:", np.array_equal(flow_o, full_g))
dg = np.abs(full_g - flow_o); do = np.abs(full_o - flow_o)
print("gpu full vs reassembled: differ", int((dg > 0).sum()), "max", float(dg.max()), "; oracle full vs reassembled: differ", int((do > 0).sum()), "max", float(do.max()))
