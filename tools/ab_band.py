"""A/B of the two SOR kernels: bit-identity of the increments and of the end-to-end flow, then timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flowreg3d_amd as fr
from flowreg3d_amd import _lib
from flowreg3d_amd.synthetic import make_pair

_lib.init(0)
rng = np.random.default_rng(0)


def run(kernel, fn):
    os.environ["FR3D_SOR_KERNEL"] = str(kernel)
    return fn()


for shape, C, iters, lag in [((9, 17, 21), 1, 7, 3), ((24, 40, 72), 2, 12, 5), ((70, 66, 130), 1, 20, 5),
                             ((16, 200, 300), 1, 9, 4), ((5, 5, 5), 1, 3, 1), ((33, 64, 64), 3, 1, 1)]:
    fixed, moving, _ = make_pair(shape, seed=3, channels=C)
    kw = dict(alpha=(0.25, 0.3, 0.2), update_lag=lag, iterations=iters, min_level=0, levels=3, eta=0.8,
              a_smooth=1.0, a_data=0.45)
    f0 = run(0, lambda: fr.get_displacement(fixed, moving, **kw))
    f1 = run(1, lambda: fr.get_displacement(fixed, moving, **kw))
    print(shape, C, iters, "max|diff| = %.3e" % np.abs(f0 - f1).max(), "identical" if np.array_equal(f0, f1) else "DIFFERENT",
          flush=True)

if len(sys.argv) > 1:
    n = int(sys.argv[1])
    fixed, moving, _ = make_pair((n, n, n), seed=1)
    kw = dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=100, min_level=0, levels=5, eta=0.8, a_smooth=1.0,
              a_data=0.45)
    for kern in (0, 1, 0, 1):
        run(kern, lambda: fr.get_displacement(fixed, moving, **kw))
        _lib.load().fr3d_prof_enable(1); _lib.load().fr3d_prof_reset()
        t0 = time.time(); run(kern, lambda: fr.get_displacement(fixed, moving, **kw)); dt = time.time() - t0
        st = _lib.prof_get()["sor"]
        print("kernel", kern, "total %.1f ms; sor %.1f ms, %d launches, %.2f TB/s" %
              (dt * 1e3, st["ms"], st["launches"], st["algo_bytes"] / st["ms"] / 1e9), flush=True)
        _lib.load().fr3d_prof_enable(0)
