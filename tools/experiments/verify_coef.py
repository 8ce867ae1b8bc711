#!/usr/bin/env python3
"""spline coefficients: engine (pad-free prefilter) vs oracle (12-voxel pad, whole padded array filtered), bit for bit"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowreg3d_amd import _lib
from flowreg3d_amd.synthetic import make_pair
from oracle import oracle
oracle.build()
lib = _lib.init(0)
shape = tuple(int(v) for v in sys.argv[1].split(","))
_, moving, _ = make_pair(shape, seed=7, cheap=True)
Z, Y, X = shape
want = oracle.spline_filter3(np.pad(moving.astype(np.float64), 12, mode="edge"))[10:-10, 10:-10, 10:-10]
got = np.empty((Z + 4, Y + 4, X + 4), np.float64)
vol = np.ascontiguousarray(moving, np.float32)
_lib.check(lib.fr3d_spline_coefficients(_lib.ptr(vol), Z, Y, X, _lib.ptr(got)))
d = got != want
rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-300)
print(shape, "coefficients differ", int(d.sum()), "of", d.size, "max rel", float(rel.max()))
if d.any():
    idx = np.argwhere(d)
    print("first", idx[:5].tolist(), "z range", idx[:, 0].min(), idx[:, 0].max(), "y range", idx[:, 1].min(), idx[:, 1].max(), "x range", idx[:, 2].min(), idx[:, 2].max())
    # which axes' positions are affected
    print("fraction per x position (first 12):", [round(float(d[:, :, i].mean()), 4) for i in range(12)], "... last 6", [round(float(d[:, :, -i].mean()), 4) for i in range(1, 7)])
