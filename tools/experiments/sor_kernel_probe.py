#!/usr/bin/env python3
"""A/B timing of the SOR sweep kernels (FR3D_SOR_KERNEL = step | pair6 | pair14) on one box, interleaved.
Record of the round-2 experiment "two hyperplanes of one iteration per launch" (k_sor_pair.hip in commit
"WIP: typed raw-volume executor entry ..."): bit-identical to k_sor_step on 26 shape/mode cases, but 14-35 %
SLOWER (profiles/r02/kprobe*.jsonl: 256^3 90-97 ms against 71-79 ms per volume, 512^3 669-698 against 530-589) --
the halo row/lane recomputation eats the saved plane reads and the phase barrier halves the duty cycle of the
loads.  The kernel and the FR3D_SOR_KERNEL switch were removed again; this script needs that commit to run.
usage (GPU box): python tools/experiments/sor_kernel_probe.py EDGE BATCH [reps] [kernels,comma,separated]"""
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
    nb = int(sys.argv[2])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    kernels = sys.argv[4].split(",") if len(sys.argv) > 4 else ["step", "pair6", "pair14"]
    lib = _lib.init(0)
    levels = {64: 2, 128: 3, 256: 4, 512: 5}.get(n, 4)
    fixed, moving, _ = fast_pair((n, n, n))
    nv = n ** 3
    params = _lib.make_params(alpha=(0.25,) * 3, update_lag=5, iterations=100, min_level=0, levels=levels, eta=0.8,
                              a_smooth=1.0, a_data=0.45, n_channels=1, solver_fp64=1)
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

    for k in kernels:  # allocate every workspace first
        os.environ["FR3D_SOR_KERNEL"] = k
        run(False)
    t0 = time.time()
    while time.time() - t0 < 15:
        run(False)
    for rep in range(reps):
        for k in kernels:
            os.environ["FR3D_SOR_KERNEL"] = k
            run(False)
            t0 = time.perf_counter()
            run(True)
            wall = time.perf_counter() - t0
            s = _lib.prof_get()["sor"]
            print(json.dumps({"edge": n, "batch": nb, "kernel": k, "rep": rep, "sor_ms_per_vol": round(s["ms"] / nb, 2),
                              "algo_GBs": round(s["algo_bytes"] / s["ms"] / 1e6, 1), "frac": round(s["algo_bytes"] / s["ms"] / 8e9, 4),
                              "launches": s["launches"], "avg_launch_us": round(1e3 * s["ms"] / s["launches"], 1),
                              "wall_ms_per_vol": round(1e3 * wall / nb, 1)}), flush=True)


if __name__ == "__main__":
    main()
