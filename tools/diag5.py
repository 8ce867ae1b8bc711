import sys, numpy as np
sys.path.insert(0,'.')
import flowreg3d_amd as fr
from flowreg3d_amd import _lib
from flowreg3d_amd.synthetic import make_pair, epe
from oracle import oracle
_lib.init(0)
base = dict(alpha=(0.25,)*3, update_lag=5, iterations=60, min_level=0, levels=4, eta=0.8, a_smooth=1.0, a_data=0.45)
def run(tag, shape, C, motion, scale, **over):
    kw = dict(base); kw.update(over)
    fixed, moving, gt = make_pair(shape, seed=1234, channels=C, motion=motion, scale=scale)
    if C == 2: kw['weight'] = np.array([0.5, 0.5])
    want = oracle.get_displacement(fixed, moving, **kw)
    for fp64 in (0, 1, 2):
        got = fr.get_displacement(fixed, moving, solver_fp64=fp64, **kw)
        print(f"{tag:40s} fp64={fp64}: EPE mean {epe(got,want)[0]:.3e} max {epe(got,want)[1]:.3e} |flow|max {np.abs(want).max():.2f}", flush=True)
run("c2 expansion s1", (24,48,48), 2, "expansion", 1.0)
run("c1 expansion s1", (24,48,48), 1, "expansion", 1.0)
run("c2 expansion s0.3", (24,48,48), 2, "expansion", 0.3)
run("c2 rigid s1", (24,48,48), 2, "rigid", 1.0)
run("c1 rigid s1", (24,48,48), 1, "rigid", 1.0)
run("c2 expansion s1 levels2", (24,48,48), 2, "expansion", 1.0, levels=2)
run("c2 expansion s1 it20", (24,48,48), 2, "expansion", 1.0, iterations=20)
run("c2 expansion s1 lag1", (24,48,48), 2, "expansion", 1.0, update_lag=1)
run("c2 expansion s1 adata1", (24,48,48), 2, "expansion", 1.0, a_data=1.0)
