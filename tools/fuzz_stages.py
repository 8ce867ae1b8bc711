"""Randomised stage-level checks, GPU vs CPU oracle: resampler (bit-exact), cubic/linear warp
(<= 2 fp32 ulp of the intensity range), 5^3 median (bit-exact).
usage (GPU box): python tools/fuzz_stages.py [n_cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flowreg3d_amd as fr
from flowreg3d_amd import _lib
from oracle import oracle


def run(n_cases=40, seed=0, verbose=True):
    say = print if verbose else (lambda *a, **k: None)
    rng = np.random.default_rng(seed)
    _lib.init()
    bad = 0
    dims = [1, 2, 3, 4, 5, 6, 7, 8, 13, 20, 33, 64, 65, 90]
    for case in range(n_cases):
        shape = tuple(int(v) for v in rng.choice(dims, size=3))
        vol = rng.random(shape).astype(np.float32)
        # resampler: arbitrary target sizes, down and up
        size = tuple(int(v) for v in rng.choice(dims, size=3))
        a = fr.imresize_fused_gauss_cubic3D(vol, size)
        b = oracle.imresize_fused_gauss_cubic3D(vol, size)
        ok_r = a.shape == b.shape and np.array_equal(a, b)
        # warp: displacements from sub-voxel to far out of the volume
        mag = float(rng.choice([0.3, 2.0, 15.0, 200.0]))
        u, v, w = (mag * rng.standard_normal(shape) for _ in range(3))
        ref = rng.random(shape).astype(np.float32)
        ok_w = True
        for method in ("cubic", "linear"):
            g = np.asarray(fr.imregister_wrapper(vol, u, v, w, ref, method), np.float64).reshape(shape)
            o = np.asarray(oracle.imregister_wrapper(vol, u, v, w, ref, method), np.float64).reshape(shape)
            ok_w = ok_w and np.isfinite(g).all() and np.abs(g - o).max() <= 3e-7 * max(1.0, np.abs(o).max())
        # median (the path applies it when min(shape) > 5, the kernel itself takes any shape)
        d = rng.standard_normal(shape).astype(np.float32)
        ok_m = np.array_equal(fr.median_filter5(d), oracle.median5(d.astype(np.float64)).astype(np.float32))
        if not (ok_r and ok_w and ok_m):
            bad += 1
        say("%s case %2d shape %s -> %s  |disp| %.1f: resize %s warp %s median %s" %
            ("ok " if ok_r and ok_w and ok_m else "BAD", case, shape, size, mag, ok_r, ok_w, ok_m), flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("cases %d bad %d" % (n, run(n, sd)))
