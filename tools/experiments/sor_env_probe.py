#!/usr/bin/env python3
"""A/B timing of the SOR sweep under different values of one environment switch that the engine reads per call
(e.g. FR3D_SOR_LW = 64|32|16, lanes per row segment), interleaved on one box; also checks that every setting gives
bit-identical flows.
usage (GPU box): [FR3D_PROBE_MODE=0..3] python tools/experiments/sor_env_probe.py EDGE BATCH VAR v1,v2,... [reps]
(the switches exist in the experiment build only: FR3D_LIB=flowreg3d_amd/lib/libflowreg3d_hip_exp.so)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowreg3d_amd import _lib  # noqa: E402
from flowreg3d_amd.synthetic import fast_pair  # noqa: E402


def main():
    n, nb, var = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    values = sys.argv[4].split(",")
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    lib = _lib.init(0)
    levels = {64: 2, 128: 3, 256: 4, 512: 5}.get(n, 4)
    fixed, moving, _ = fast_pair((n, n, n))
    nv = n ** 3
    params = _lib.make_params(alpha=(0.25,) * 3, update_lag=5, iterations=100, min_level=0, levels=levels, eta=0.8,
                              a_smooth=float(os.environ.get("FR3D_PROBE_ASMOOTH", "1.0")), a_data=0.45, n_channels=1,
                              solver_fp64=int(os.environ.get("FR3D_PROBE_MODE", "1")))
    ref = lib.fr3d_dev_malloc(nv * 4)
    mov = lib.fr3d_dev_malloc(nv * 4 * nb)
    flows = lib.fr3d_dev_malloc(nv * 12 * nb)
    regs = lib.fr3d_dev_malloc(nv * 4 * nb)
    lib.fr3d_h2d(ref, fixed.ctypes.data, nv * 4)
    for b in range(nb):
        lib.fr3d_h2d(mov + b * nv * 4, moving.ctypes.data, nv * 4)
    lib.fr3d_set_batch(nb)

    def run(prof):
        lib.fr3d_prof_enable(1 if prof else 0)
        if prof:
            lib.fr3d_prof_reset()
        _lib.check(lib.fr3d_process_batch_dev(C.byref(params), mov, mov, ref, ref, None, None, nb, n, n, n, 1, 3, flows,
                                              regs, C.cast(None, _lib.PROGRESS_FN), None))
        lib.fr3d_sync()

    first = None
    for v in values:
        os.environ[var] = v
        run(False)
        out = np.empty((n, n, n, 3), np.float32)
        lib.fr3d_d2h(out.ctypes.data, flows, nv * 12)
        if first is None:
            first = out
        print(json.dumps({"var": var, "value": v, "bit_identical_to_first": bool(np.array_equal(first, out))}), flush=True)
    t0 = time.time()
    while time.time() - t0 < 12:
        run(False)
    for rep in range(reps):
        for v in values:
            os.environ[var] = v
            run(False)
            t0 = time.perf_counter()
            run(True)
            wall = time.perf_counter() - t0
            s = _lib.prof_get()["sor"]
            print(json.dumps({"edge": n, "batch": nb, "mode": int(os.environ.get("FR3D_PROBE_MODE", "1")), var: v, "rep": rep, "sor_ms_per_vol": round(s["ms"] / nb, 2),
                              "frac": round(s["algo_bytes"] / s["ms"] / 8e9, 4), "launches": s["launches"],
                              "wall_ms_per_vol": round(1e3 * wall / nb, 1)}), flush=True)


if __name__ == "__main__":
    main()
