#!/usr/bin/env python3
"""fp64 motion tensor: engine (fr3d_motion_tensor_f64) vs oracle, bit for bit, per entry"""
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
h = tuple(float(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1.0, 1.0, 1.0)
fixed, moving, _ = make_pair(shape, seed=7, cheap=True)
Z, Y, X = shape
J = oracle.get_motion_tensor_gc(fixed, moving, *h)
want = np.stack([j[1:-1, 1:-1, 1:-1] for j in J])
got = np.empty((10, Z, Y, X), np.float64)
_lib.check(lib.fr3d_motion_tensor_f64(_lib.ptr(np.ascontiguousarray(fixed, np.float32)), _lib.ptr(np.ascontiguousarray(moving, np.float32)), Z, Y, X, *h, _lib.ptr(got)))
names = "J11,J22,J33,J44,J12,J13,J23,J14,J24,J34".split(",")
for q in range(10):
    d = got[q] != want[q]
    if d.any():
        i = tuple(np.argwhere(d)[0])
        print(names[q], "differ", int(d.sum()), "first", i, repr(got[q][i]), repr(want[q][i]))
print("total differing entries", int((got != want).sum()), "of", got.size)
