#!/usr/bin/env python3
"""Rate of the flow path on shapes a microscopy user brings (thin stacks, small cubes) against the 256^3 benchmark
shape: voxel updates per second of a lock-step batch through fr3d_process_batch_dev (device-resident, two lanes), and
of single get_displacement calls.  Same solver parameters as bench.py (100 iterations, lag 5, eta 0.8).
   tools/experiments/shape_rate_probe.py [Z,Y,X ...]"""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from flowreg3d_amd import _lib, core
from flowreg3d_amd.synthetic import fast_pair

shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or \
    [(256, 256, 256), (32, 512, 512), (16, 1024, 1024), (64, 256, 256), (100, 100, 100), (48, 48, 48), (8, 2048, 2048)]
lib = _lib.init(0)
for shape in shapes:
    Z, Y, X = shape
    nv = Z * Y * X
    fixed, moving, _ = fast_pair(shape, block=min(64, min(shape)))
    kw = bench.solver_kwargs(50)
    sizes, _ = core.pyramid_schedule(Z, Y, X, kw["eta"], kw["levels"], kw["min_level"])
    updates = sum(z * y * x for z, y, x in sizes) * kw["iterations"]
    params = _lib.make_params(n_channels=1, **kw)
    n = max(2, min(int(os.environ.get("PROBE_NMAX", "16")), (1 << 27) // nv))
    if os.environ.get("PROBE_SET_BATCH"):
        lib.fr3d_set_batch(min(n, int(os.environ["PROBE_SET_BATCH"])))
    dfix = bench.DevArray(lib, (Z, Y, X, 1)).upload(fixed)
    dbat = bench.DevArray(lib, (n, Z, Y, X, 1)).upload(np.broadcast_to(moving, (n,) + shape).copy())
    flows = bench.DevArray(lib, (n, Z, Y, X, 3))
    regs = bench.DevArray(lib, (n, Z, Y, X, 1))
    best = None
    for rep in range(3):
        lib.fr3d_sync()
        t0 = time.perf_counter()
        _lib.check(lib.fr3d_process_batch_dev(C.byref(params), dbat.ptr, dbat.ptr, dfix.ptr, dfix.ptr, None, None, n, Z, Y, X, 1,
                                              3, flows.ptr, regs.ptr, _lib.PROGRESS_FN(0), None))
        lib.fr3d_sync()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    dmov = bench.DevArray(lib, (Z, Y, X, 1)).upload(moving)
    dflow = bench.DevArray(lib, (Z, Y, X, 3))
    one = None
    for rep in range(3):
        lib.fr3d_sync()
        t0 = time.perf_counter()
        _lib.check(lib.fr3d_get_displacement_dev(C.byref(params), dfix.ptr, dmov.ptr, Z, Y, X, 1, None, None, dflow.ptr))
        lib.fr3d_sync()
        dt = time.perf_counter() - t0
        one = dt if one is None or dt < one else one
    print(json.dumps({"shape": shape, "levels": len(sizes), "batch": n, "volumes_per_s": round(n / best, 2),
                      "Gupdates_per_s_batch": round(updates * n / best / 1e9, 2), "single_call_ms": round(one * 1e3, 2),
                      "Gupdates_per_s_single": round(updates / one / 1e9, 2), "solver_mode": lib.fr3d_last_solver_mode()}), flush=True)
    for a in (dfix, dbat, flows, regs, dmov, dflow):
        a.free()
