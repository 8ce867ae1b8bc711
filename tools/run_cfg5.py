#!/usr/bin/env python3
"""BASELINE config 5 on the MI355X: 512x512x256 (X,Y,Z) two-channel pair, expansion/contraction +
rotation, levels=8/min_level=0 (9 solves), weights 0.5/0.5; flow end-point error GPU vs the CPU
oracle.  The oracle needs minutes per volume at full size, so the comparison volume is
configurable (--scale 0.5 = 128x256x256 by default); the GPU also runs the full-size case and
reports its time and its EPE against the synthetic ground truth.

usage (GPU box): python tools/run_cfg5.py [--scale 0.5] [--full] > gpurun_out/cfg5.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.5)
    ap.add_argument("--full", action="store_true", help="also run the full 256x512x512 case on the GPU")
    ap.add_argument("--iterations", type=int, default=100)
    args = ap.parse_args()
    import flowreg3d_amd as fr
    from flowreg3d_amd import _lib
    from flowreg3d_amd.synthetic import epe, make_pair
    from oracle import oracle
    _lib.init(0)
    kw = dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=args.iterations, min_level=0, levels=8, eta=0.8,
              a_smooth=1.0, a_data=0.45, weight=np.array([0.5, 0.5]))
    out = {"config": "cfg5: 512x512x256 (X,Y,Z), C=2, expansion+rotation, levels=8, min_level=0", "params": {
        k: (list(map(float, v)) if hasattr(v, "__len__") else v) for k, v in kw.items()}}

    shape = tuple(int(round(s * args.scale)) for s in (256, 512, 512))
    fixed, moving, gt = make_pair(shape, seed=1234, channels=2, motion="expansion", cheap=True)
    t0 = time.perf_counter()
    got = fr.get_displacement(fixed, moving, solver_fp64=0, **kw)  # fp32 solver storage (fast mode)
    t_gpu = time.perf_counter() - t0
    got64 = fr.get_displacement(fixed, moving, **kw)  # default for C >= 2: fp64 solver storage
    t0 = time.perf_counter()
    want = oracle.get_displacement(fixed, moving, **kw)
    t_cpu = time.perf_counter() - t0
    crop = 8 if min(shape) >= 64 else 4
    out["parity"] = {"shape_zyx": shape, "levels_solved": len(fr.pyramid_schedule(*shape, 0.8, 8, 0)[0]),
                     "epe_gpu_vs_cpu_mean": epe(got64, want)[0], "epe_gpu_vs_cpu_max": epe(got64, want)[1],
                     "epe_gpu_vs_cpu_mean_interior": epe(got64, want, crop)[0],
                     "epe_gpu_fp32_storage_vs_cpu_mean": epe(got, want)[0],
                     "epe_gpu_fp32_storage_vs_cpu_max": epe(got, want)[1],
                     "epe_gpu_vs_gt_mean_interior": epe(got64, gt, crop)[0],
                     "epe_cpu_vs_gt_mean_interior": epe(want, gt, crop)[0],
                     "gpu_fp32_storage_seconds_incl_pcie": t_gpu, "cpu_seconds_1core": t_cpu}
    if args.full:
        shape = (256, 512, 512)
        fixed, moving, gt = make_pair(shape, seed=1234, channels=2, motion="expansion", cheap=True)
        fr.get_displacement(fixed, moving, **kw)  # warm-up (allocations, tables)
        t0 = time.perf_counter()
        got = fr.get_displacement(fixed, moving, **kw)
        t_gpu = time.perf_counter() - t0
        out["full_size_gpu"] = {"shape_zyx": shape, "levels_solved": len(fr.pyramid_schedule(*shape, 0.8, 8, 0)[0]),
                                "gpu_seconds_incl_pcie": t_gpu,
                                "epe_gpu_vs_gt_mean_interior": epe(got, gt, 8)[0],
                                "flow_mean": [float(x) for x in got.mean(axis=(0, 1, 2))]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
