"""Window sweep against plane sweep: bit-identity and time of one get_displacement call (GPU box).
   python tools/experiments/win_probe.py [size] [modes]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import flowreg3d_amd as fr
from flowreg3d_amd import _lib
from flowreg3d_amd.synthetic import make_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
modes = [int(m) for m in (sys.argv[2] if len(sys.argv) > 2 else "2,3,1").split(",")]
levels = {64: 2, 128: 3, 256: 4, 512: 5}.get(n, 3)
fixed, moving, gt = make_pair((n, n, n), seed=1234, cheap=True)
_lib.init(0)
lib = _lib.load()
for mode in modes:
    res = {}
    for sweep in (1, 2):
        kw = dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=100, levels=levels, eta=0.8, a_smooth=1.0, a_data=0.45,
                  solver_fp64=mode, solver_sweep=sweep)
        fr.get_displacement(fixed, moving, **kw)  # warm-up: allocations, schedules
        lib.fr3d_prof_enable(1); lib.fr3d_prof_reset()
        t = time.time()
        flow = fr.get_displacement(fixed, moving, **kw)
        dt = time.time() - t
        st = _lib.prof_get()
        lib.fr3d_prof_enable(0)
        res[sweep] = flow
        print(f"n={n} mode={mode} sweep={sweep}: call {dt*1e3:8.1f} ms, sor {st['sor']['ms']:8.2f} ms in {st['sor']['launches']} launches, "
              f"epe_gt {np.linalg.norm(flow - gt, axis=-1)[8:-8,8:-8,8:-8].mean():.4f}", flush=True)
    same = np.array_equal(res[1], res[2])
    print(f"n={n} mode={mode}: bit-identical {same}" + ("" if same else f" max diff {np.abs(res[1]-res[2]).max():.3e}"), flush=True)
