"""PCIe-inclusive rate of the plugin entry point: HipExecutor3D.process_batch on host (NumPy) arrays."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowreg3d_amd.executor import HipExecutor3D
from flowreg3d_amd.synthetic import fast_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
fixed, moving, _ = fast_pair((n, n, n))
batch = np.stack([moving] * T)[..., None]
fp = dict(alpha=(0.25,) * 3, update_lag=5, iterations=100, min_level=0, levels=4 if n < 512 else 5, eta=0.8,
          a_smooth=1.0, a_data=0.45)
w0 = np.zeros(fixed.shape + (3,), np.float32)
with HipExecutor3D() as ex:
    for rep in range(3):
        t0 = time.perf_counter()
        reg, flows = ex.process_batch(batch, batch, fixed[..., None], fixed[..., None], w0, None, None, flow_params=fp)
        dt = time.perf_counter() - t0
        print("rep %d: %d volumes of %d^3 in %.3f s = %.2f volumes/s (host arrays in, host arrays out)" %
              (rep, T, n, dt, T / dt), flush=True)
