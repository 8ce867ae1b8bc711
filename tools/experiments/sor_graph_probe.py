#!/usr/bin/env python3
"""Experiment: SOR sweep time per volume against the lock-step batch, eager launches vs one hipGraph per level
(FR3D_SOR_GRAPH=1, read once per process -> one process per setting).
usage (GPU box): FR3D_SOR_GRAPH=0|1 python tools/experiments/sor_graph_probe.py EDGE"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowreg3d_amd import _lib  # noqa: E402
from flowreg3d_amd.synthetic import fast_pair  # noqa: E402


def main():
    n = int(sys.argv[1])
    lib = _lib.init(0)
    levels = {128: 3, 256: 4, 512: 5}[n]
    batches = (1, 2, 4, 8) if n <= 256 else (1, 2, 4)
    fixed, moving, _ = fast_pair((n, n, n))
    nv = n ** 3
    bmax = max(batches)
    params = _lib.make_params(alpha=(0.25,) * 3, update_lag=5, iterations=100, min_level=0, levels=levels, eta=0.8,
                              a_smooth=1.0, a_data=0.45, n_channels=1, solver_fp64=1)
    ref = lib.fr3d_dev_malloc(nv * 4)
    mov = lib.fr3d_dev_malloc(nv * 4 * bmax)
    flows = lib.fr3d_dev_malloc(nv * 12 * bmax)
    regs = lib.fr3d_dev_malloc(nv * 4 * bmax)
    lib.fr3d_h2d(ref, fixed.ctypes.data, nv * 4)
    for b in range(bmax):
        lib.fr3d_h2d(mov + b * nv * 4, moving.ctypes.data, nv * 4)

    def run(nb, prof):
        lib.fr3d_prof_enable(1 if prof else 0)
        if prof:
            lib.fr3d_prof_reset()
        _lib.check(lib.fr3d_process_batch_dev(C.byref(params), mov, mov, ref, ref, None, None, nb, n, n, n, 1, 3, flows,
                                              regs, C.cast(None, _lib.PROGRESS_FN), None))
        lib.fr3d_sync()

    lib.fr3d_set_batch(bmax)
    t0 = time.time()
    while time.time() - t0 < 12:
        run(bmax, False)
    for rep in range(2):
        for nb in batches:
            lib.fr3d_set_batch(nb)
            run(nb, False)
            run(nb, False)
            t0 = time.perf_counter()
            run(nb, True)
            wall = time.perf_counter() - t0
            s = _lib.prof_get()["sor"]
            print(json.dumps({"edge": n, "graph": os.environ.get("FR3D_SOR_GRAPH", "0"), "batch": nb, "rep": rep,
                              "sor_ms_per_vol": round(s["ms"] / nb, 2), "frac": round(s["algo_bytes"] / s["ms"] / 8e9, 4),
                              "launches": s["launches"], "wall_ms_per_vol": round(1e3 * wall / nb, 1)}), flush=True)


if __name__ == "__main__":
    main()
