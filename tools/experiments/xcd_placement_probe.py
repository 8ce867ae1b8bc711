#!/usr/bin/env python3
"""Which XCD do the workgroups of a launch run on (fr3d_xcd_probe: HW_REG_XCC_ID per workgroup)?  The sweep's XCD-aware tile
order assumes round-robin placement: ids with equal blockIdx.x % 8 share an XCD.  Prints, for several grid shapes, the
XCC ids of the first workgroups, how often b and b + 8 share an XCD, how the label of blockIdx.x = 0 moves from one
volume (blockIdx.y) to the next, and -- for the box-state question -- the sweep's rate in the same process.
usage (GPU box): python tools/experiments/xcd_placement_probe.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowreg3d_amd import _lib  # noqa: E402

lib = _lib.init(0)
out = {}
for gx, gy in ((64, 1), (4096, 1), (4099, 1), (4096, 8), (4099, 8), (7001, 4)):
    a = np.full((gy, gx), -1, np.int32)
    _lib.check(lib.fr3d_xcd_probe(gx, gy, a.ctypes.data))
    lin = a.reshape(-1)
    same8 = float(np.mean(a[:, 8:] == a[:, :-8]))
    rr = float(np.mean((lin[1:] - lin[:-1]) % 8 == 1))
    out[f"{gx}x{gy}"] = {"first16": a[0, :16].tolist(), "b_and_b+8_share_an_xcd": round(same8, 4),
                         "linear_id_plus_1_is_next_xcd": round(rr, 4), "x0_of_each_volume": a[:, 0].tolist(),
                         "workgroups_per_xcd": np.bincount(lin, minlength=8).tolist()}
st = C.c_double(0.0)
lib.fr3d_stream_probe(1 << 28, 10, C.byref(st))
out["stream_GBs"] = round(st.value)
print(json.dumps(out))
