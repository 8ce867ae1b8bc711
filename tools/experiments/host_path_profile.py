import cProfile, pstats, time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from flowreg3d_amd.executor import HipExecutor3D
from flowreg3d_amd.synthetic import fast_pair
import bench
Z=Y=X=256
fixed, moving, _ = fast_pair((Z, Y, X))
batch = np.ascontiguousarray(np.stack([moving] * 8)[..., None])
fp = dict(bench.solver_kwargs(4), weight=np.array([1.0]), solver_fp64=None)
w0 = np.zeros((Z, Y, X, 3), np.float32)
ref = fixed[..., None]
with HipExecutor3D() as ex:
    ex.process_batch(batch, batch, ref, ref, w0, None, None, flow_params=fp)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    reg, flows = ex.process_batch(batch, batch, ref, ref, w0, None, None, flow_params=fp)
    pr.disable()
    print("total", time.perf_counter() - t0)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
