#!/usr/bin/env python3
"""Round-2 experiment record: the cubic warp gather with the coefficient box of a 32x8x4 output tile staged in
LDS (k_warp_cubic_lds, removed again) against the one-thread-per-voxel global gather.  Result on MI355X:
the LDS variant was SLOWER in every form tried (runtime strides 0.69 ms, compile-time strides 0.69 ms, unrolled
staging loads 0.67 ms per 256^3 warp against 0.53 ms) -- the gather is bound by the 256 fp64 operations per
voxel and the tap addressing, not by where the coefficients come from.  What did pay: dropping the per-tap
clamps (coordinates are clipped and the grid is padded by 12, so the 4^3 taps never leave it): four x-taps of a
row become consecutive loads behind one address, 0.53 -> 0.37 ms at 256^3, 4.55 -> 3.23 ms at 512^3,
bit-identical.  The FR3D_WARP switch no longer exists; the script now just times the shipped kernel twice.
usage (GPU box): python tools/experiments/warp_probe.py [edge]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowreg3d_amd import _lib  # noqa: E402
from flowreg3d_amd.synthetic import fast_pair, flow_gt  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    lib = _lib.init(0)
    fixed, moving, _ = fast_pair((n, n, n))
    nv = n ** 3
    vol = lib.fr3d_dev_malloc(nv * 4)
    flow = lib.fr3d_dev_malloc(nv * 12)
    out = lib.fr3d_dev_malloc(nv * 4)
    lib.fr3d_h2d(vol, moving.ctypes.data, nv * 4)
    g = np.ascontiguousarray(flow_gt((n, n, n)))
    lib.fr3d_h2d(flow, g.ctypes.data, nv * 12)
    for rep in range(3):
        for mode in ("global", "lds"):
            if mode == "global":
                os.environ["FR3D_WARP"] = "global"
            else:
                os.environ.pop("FR3D_WARP", None)
            for _ in range(2):
                _lib.check(lib.fr3d_warp_dev(vol, _lib.F32, flow, _lib.F32, vol, n, n, n, 1, 3, out))
            lib.fr3d_prof_enable(1)
            lib.fr3d_prof_reset()
            for _ in range(5):
                _lib.check(lib.fr3d_warp_dev(vol, _lib.F32, flow, _lib.F32, vol, n, n, n, 1, 3, out))
            st = _lib.prof_get()
            lib.fr3d_prof_enable(0)
            print(json.dumps({"edge": n, "mode": mode, "rep": rep, "warp_ms": round(st["warp"]["ms"] / 5, 4),
                              "warp_GBs_algo": round(st["warp"]["algo_bytes"] / st["warp"]["ms"] / 1e6, 1),
                              "prefilter_ms": round(st["prefilter"]["ms"] / 5, 4)}), flush=True)


if __name__ == "__main__":
    main()
