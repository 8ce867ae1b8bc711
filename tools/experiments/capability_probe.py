#!/usr/bin/env python3
"""How large a volume / lock-step batch fits since the solver slabs are compact (round 2)?  Runs one
fr3d_process_batch_dev call per setting on synthetic volumes and prints time, batch actually used and peak HBM.
usage (GPU box): python tools/experiments/capability_probe.py EDGE BATCH MODE [levels]"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowreg3d_amd import _lib  # noqa: E402


def hbm_used_gib():
    try:
        out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        k = next(iter(d))
        return int(d[k]["VRAM Total Used Memory (B)"]) / 2 ** 30
    except Exception:
        return None


def main():
    n, nb, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    levels = int(sys.argv[4]) if len(sys.argv) > 4 else {256: 4, 512: 5, 768: 6, 1024: 6}.get(n, 4)
    lib = _lib.init(0)
    nv = n ** 3
    rng = np.random.default_rng(0)
    base = rng.random((64, 64, 64), dtype=np.float32)
    reps = -(-n // 64)
    fixed = np.tile(base, (reps, reps, reps))[:n, :n, :n].copy()
    from scipy.ndimage import uniform_filter
    fixed = uniform_filter(fixed, 5, mode="wrap").astype(np.float32)
    moving = np.roll(fixed, 1, axis=2)
    params = _lib.make_params(alpha=(0.25,) * 3, update_lag=5, iterations=100, min_level=0, levels=levels, eta=0.8,
                              a_smooth=1.0, a_data=0.45, n_channels=1, solver_fp64=mode)
    ref = lib.fr3d_dev_malloc(nv * 4)
    mov = lib.fr3d_dev_malloc(nv * 4 * nb)
    flows = lib.fr3d_dev_malloc(nv * 12 * nb)
    regs = lib.fr3d_dev_malloc(nv * 4 * nb)
    assert ref and mov and flows and regs, "device allocation of the series failed"
    lib.fr3d_h2d(ref, fixed.ctypes.data, nv * 4)
    for b in range(nb):
        lib.fr3d_h2d(mov + b * nv * 4, moving.ctypes.data, nv * 4)
    lib.fr3d_set_batch(nb)
    res = {"edge": n, "batch_asked": nb, "mode": mode, "levels": levels}
    for rep in range(2):
        lib.fr3d_prof_enable(1)
        lib.fr3d_prof_reset()
        t0 = time.perf_counter()
        rc = lib.fr3d_process_batch_dev(C.byref(params), mov, mov, ref, ref, None, None, nb, n, n, n, 1, 3, flows, regs,
                                        C.cast(None, _lib.PROGRESS_FN), None)
        lib.fr3d_sync()
        dt = time.perf_counter() - t0
        if rc != 0:
            res["error"] = _lib.last_error()[:200]
            break
        s = _lib.prof_get()["sor"]
        res.update(ms_per_vol=round(1e3 * dt / nb, 1), sor_ms_per_vol=round(s["ms"] / nb, 1),
                   sor_frac=round(s["algo_bytes"] / s["ms"] / 8e9, 4), sor_launches=s["launches"], hbm_used_gib=hbm_used_gib())
    out = np.empty((4, 4, 4, 3), np.float32)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
