#!/usr/bin/env python3
"""Flow parity against the committed full-size CPU samples (tests/golden/fullsize_*.npz) and SOR time for each
solver mode (fr3d_params.solver_fp64 = 0..4).  usage (GPU box): python tools/experiments/mode_parity_probe.py cfg2|cfg3 [modes]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import flowreg3d_amd as fr  # noqa: E402
from flowreg3d_amd import _lib  # noqa: E402
from flowreg3d_amd.synthetic import fullsize_case  # noqa: E402


def epe(a, b):
    d = np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64), axis=-1)
    return float(d.mean()), float(d.max())


def main():
    case = sys.argv[1]
    modes = [int(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 3, 4, 0, 2]
    lib = _lib.init(0)
    g = np.load(os.path.join(ROOT, "tests", "golden", f"fullsize_{case}.npz"))
    meta = json.loads(bytes(g["meta"]).decode())
    fixed, moving, gt, kw = fullsize_case(case, warp=fr.imregister_wrapper)
    st, bl = meta["stride"], meta["block"]
    z0, y0, x0 = meta["block_origin_zyx"]
    for m in modes:
        try:
            fr.get_displacement(fixed, moving, solver_fp64=m, **kw)  # warm-up / allocation
            lib.fr3d_prof_enable(1)
            lib.fr3d_prof_reset()
            t0 = time.perf_counter()
            flow = fr.get_displacement(fixed, moving, solver_fp64=m, **kw)
            dt = time.perf_counter() - t0
            s = _lib.prof_get()["sor"]
            lib.fr3d_prof_enable(0)
        except RuntimeError as e:
            print(json.dumps({"case": case, "mode": m, "error": str(e)[:200]}), flush=True)
            continue
        lm, lx = epe(flow[::st, ::st, ::st], g["lattice"])
        bm, bx = epe(flow[z0:z0 + bl, y0:y0 + bl, x0:x0 + bl], g["block"])
        print(json.dumps({"case": case, "mode": m, "lattice_mean": lm, "lattice_max": lx, "block_mean": bm, "block_max": bx,
                          "sor_ms": round(s["ms"], 1), "sor_frac_batch1": round(s["algo_bytes"] / s["ms"] / 8e9, 4),
                          "call_s": round(dt, 2)}), flush=True)


if __name__ == "__main__":
    main()
