"""One-off: BASELINE config 2 at full size (256^3, 5 levels, 100 iterations) -- GPU flow vs the CPU oracle.
The oracle needs a few minutes on one core.  usage (GPU box): python tools/parity_fullsize.py [edge]"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flowreg3d_amd as fr
from flowreg3d_amd.synthetic import fast_pair, epe
from oracle import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fixed, moving, gt = fast_pair((n, n, n))
kw = dict(alpha=(0.25,) * 3, update_lag=5, iterations=100, min_level=0, levels=4, eta=0.8, a_smooth=1.0, a_data=0.45)
t0 = time.time(); got = fr.get_displacement(fixed, moving, solver_fp64=0, **kw); t_gpu = time.time() - t0
got64 = fr.get_displacement(fixed, moving, solver_fp64=2, **kw)
got1 = fr.get_displacement(fixed, moving, solver_fp64=1, **kw)  # fp64 arithmetic, fp32 storage
print("gpu done %.2f s; running the oracle ..." % t_gpu, flush=True)
t0 = time.time(); want = oracle.get_displacement(fixed, moving, **kw); t_cpu = time.time() - t0
out = {"shape": [n, n, n], "levels": 5, "iterations": 100,
       "epe_fp32_storage_mean": epe(got, want)[0], "epe_fp32_storage_max": epe(got, want)[1],
       "epe_fp32_storage_mean_interior8": epe(got, want, 8)[0],
       "epe_fp64_arithmetic_fp32_storage_mean": epe(got1, want)[0], "epe_fp64_arithmetic_fp32_storage_max": epe(got1, want)[1],
       "epe_fp64_storage_mean": epe(got64, want)[0], "epe_fp64_storage_max": epe(got64, want)[1],
       "epe_gpu_vs_ground_truth_interior24": epe(got, gt, 24)[0], "epe_cpu_vs_ground_truth_interior24": epe(want, gt, 24)[0],
       "gpu_seconds_incl_pcie_first_call": t_gpu, "cpu_seconds_1core": t_cpu}
print(json.dumps(out))
