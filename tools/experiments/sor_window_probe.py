#!/usr/bin/env python3
"""Experiment: SOR sweep time against the number of iterations in flight (FR3D_SOR_WINDOW) and the
lock-step batch -- does keeping the in-flight hyperplanes inside the 256 MiB Infinity Cache pay?
Result (profiles/r02/window_*.jsonl): no gain -- every launch costs ~5 us of fixed overhead, so windows of
5-20 iterations lose 2-70 % and the best case (512^3, batch 4, window 20) wins 1.5 %; the engine's
FR3D_SOR_WINDOW switch was removed again after this measurement (the script is kept as the record of
how the numbers were taken; SorArgsT::t_base is what is left of it).
usage (GPU box): python tools/experiments/sor_window_probe.py [edge ...] > gpurun_out/window_probe.jsonl"""
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
    edges = [int(a) for a in sys.argv[1:]] or [256, 512]
    lib = _lib.init(0)
    for n in edges:
        levels = {128: 3, 256: 4, 512: 5}.get(n, 4)
        batches = (1, 2, 4, 8) if n <= 256 else (1, 2, 4)
        fixed, moving, _ = fast_pair((n, n, n))
        nv = n ** 3
        bmax = max(batches)
        params = _lib.make_params(alpha=(0.25,) * 3, update_lag=5, iterations=100, min_level=0, levels=levels,
                                  eta=0.8, a_smooth=1.0, a_data=0.45, n_channels=1, solver_fp64=1)
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
            _lib.check(lib.fr3d_process_batch_dev(C.byref(params), mov, mov, ref, ref, None, None, nb, n, n, n, 1, 3,
                                                  flows, regs, C.cast(None, _lib.PROGRESS_FN), None))
            lib.fr3d_sync()

        # condition the device (an idle MI355X needs ~15 s of load to reach its steady memory rate)
        lib.fr3d_set_batch(bmax)
        t0 = time.time()
        while time.time() - t0 < (15 if n <= 256 else 20):
            run(bmax, False)
        for nb in batches:
            lib.fr3d_set_batch(nb)
            for win in (0, 5, 10, 20, 0):
                if win:
                    os.environ["FR3D_SOR_WINDOW"] = str(win)
                else:
                    os.environ.pop("FR3D_SOR_WINDOW", None)
                run(nb, False)
                t0 = time.perf_counter()
                run(nb, True)
                wall = time.perf_counter() - t0
                s = _lib.prof_get()["sor"]
                print(json.dumps({"edge": n, "batch": nb, "window": win, "sor_ms_per_vol": s["ms"] / nb,
                                  "algo_GBs": s["algo_bytes"] / s["ms"] / 1e6, "launches": s["launches"],
                                  "avg_launch_us": 1e3 * s["ms"] / s["launches"], "wall_ms_per_vol": 1e3 * wall / nb}),
                      flush=True)
        for p in (ref, mov, flows, regs):
            lib.fr3d_dev_free(p)
        _lib.shutdown()
        lib = _lib.init(0)


if __name__ == "__main__":
    main()
