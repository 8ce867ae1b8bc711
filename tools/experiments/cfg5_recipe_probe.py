#!/usr/bin/env python3
"""BASELINE config 5 at full size (256x512x512, two channels): which pyramid depth / motion amplitude lets
the coarse-to-fine scheme capture the synthetic expansion + rotation?  Runs the GPU path only (the CPU
path fails in the same way on the round-1 recipe: EPE vs ground truth 3.45 voxels on both, see
tests/golden/fullsize_cfg5*.npz metadata) and prints the end-point error against the ground truth.
usage (GPU box): python tools/experiments/cfg5_recipe_probe.py > gpurun_out/cfg5_recipes.jsonl"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import flowreg3d_amd as fr  # noqa: E402
from flowreg3d_amd import _lib  # noqa: E402
from flowreg3d_amd.synthetic import SOLVER_DEFAULTS, backward_warp, epe, flow_expansion_rotation, texture  # noqa: E402


def main():
    _lib.init(0)
    shape = (256, 512, 512)
    tex = [texture(shape, 1234 + c) for c in range(2)]
    fixed = np.stack(tex, -1)
    for scale in (1.0, 0.5):
        gt = flow_expansion_rotation(shape, scale=scale)
        moving = np.stack([backward_warp(t, -gt, order=1) for t in tex], -1)
        for levels in (8, 10, 12, 14):
            kw = dict(SOLVER_DEFAULTS, levels=levels, weight=np.array([0.5, 0.5]))
            t0 = time.time()
            flow = fr.get_displacement(fixed, moving, **kw)
            dt = time.time() - t0
            d = np.linalg.norm(flow[8:-8, 8:-8, 8:-8] - gt[8:-8, 8:-8, 8:-8], axis=-1)
            print(json.dumps({"scale": scale, "levels": levels, "solves": len(fr.pyramid_schedule(*shape, 0.8, levels, 0)[0]),
                              "coarsest": fr.pyramid_schedule(*shape, 0.8, levels, 0)[0][0],
                              "gt_max": float(np.linalg.norm(gt, axis=-1).max()), "epe_mean_int8": float(d.mean()),
                              "epe_p50": float(np.percentile(d, 50)), "epe_p99": float(np.percentile(d, 99)),
                              "epe_max": float(d.max()), "gpu_s": round(dt, 2)}), flush=True)


if __name__ == "__main__":
    main()
