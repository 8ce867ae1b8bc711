import ctypes as C, sys, os, time
import numpy as np
"""Defect-correction (d = d0 + delta) form of the SOR sweep against the direct form, per storage format
(CPU model, one level; build: gcc -O2 -shared -fPIC -ffp-contract=off -o /tmp/probe_sor.so tools/numerics/probe_sor.c -lm).
SHAPE=z,y,x SCALE=motion python tools/numerics/probe_sor_dc.py"""
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
exec(open(os.path.join(HERE, "probe_sor.py")).read().split('if __name__')[0])
def run_dc(A, wt, L, dims, ax, iters, lag, adc, fM, fR, fD, fD0, fA, fL):
    Z, Y, X = dims
    out = [np.zeros(dims) for _ in range(3)]
    arr = lambda lst: (dp*len(lst))(*[a.ctypes.data_as(dp) for a in lst])
    lib.sor_probe_dc(arr(A), wt.ctypes.data_as(dp), arr(L), Z, Y, X, C.c_double(ax), C.c_double(ax), C.c_double(ax),
                  iters, lag, C.c_double(adc), fM, fR, fD, fD0, fA, fL, arr(out))
    return np.stack(out, -1)
shape = tuple(int(v) for v in os.environ.get("SHAPE", "32,64,64").split(","))
f1, f2, gt = make_pair(shape, seed=1234, scale=float(os.environ.get("SCALE", "0.3")))
f1 = f1.astype(np.float64); f2 = f2.astype(np.float64)
Jp = o.get_motion_tensor_gc(f1, f2, 1.0, 1.0, 1.0)
J = [np.ascontiguousarray(j[1:-1, 1:-1, 1:-1]) for j in Jp]
A = factors(f1, f2, 1.0, 1.0, 1.0)
wt = np.ones(shape); L = [np.zeros(shape) for _ in range(3)]
iters = 100
t=time.time()
ref = run(J, A, wt, L, shape, 0.25, iters, 5, 0.45, 8|64)  # fp64, psi and J from factors
print("direct fp64 (factors) done", time.time()-t, "|d| mean", np.linalg.norm(ref,axis=-1).mean())
def rep(name, r):
    e = np.linalg.norm(r-ref, axis=-1); print(f"{name:60s} EPE mean {e.mean():.3e} max {e.max():.3e}")
rep("direct: J32(factors fp32)+d32+wpsi32  [flags 1|8|64|2|32]", run(J, A, wt, L, shape, 0.25, iters, 5, 0.45, 1|8|64|2|32))
rep("direct: d32 only", run(J, A, wt, L, shape, 0.25, iters, 5, 0.45, 8|64|2))
rep("direct: factors fp32 only", run(J, A, wt, L, shape, 0.25, iters, 5, 0.45, 8|64|1))
rep("dc all fp64", run_dc(A, wt, L, shape, 0.25, iters, 5, 0.45, 0,0,0,0,0,0))
rep("dc M32 r32 d32, d0 f64, A f64", run_dc(A, wt, L, shape, 0.25, iters, 5, 0.45, 1,1,1,0,0,0))
rep("dc M32 r32 d32, d0 f64, A pk42", run_dc(A, wt, L, shape, 0.25, iters, 5, 0.45, 1,1,1,0,2,1))
rep("dc M32 r32 d32, d0 f64, A f32", run_dc(A, wt, L, shape, 0.25, iters, 5, 0.45, 1,1,1,0,1,1))
rep("dc M32 r32 d32, d0 pk42, A pk42", run_dc(A, wt, L, shape, 0.25, iters, 5, 0.45, 1,1,1,2,2,1))
rep("dc M32 only", run_dc(A, wt, L, shape, 0.25, iters, 5, 0.45, 1,0,0,0,0,0))
rep("dc r32 only", run_dc(A, wt, L, shape, 0.25, iters, 5, 0.45, 0,1,0,0,0,0))
rep("dc d32 only", run_dc(A, wt, L, shape, 0.25, iters, 5, 0.45, 0,0,1,0,0,0))
rep("dc all pk42 (like mode 3 but dc)", run_dc(A, wt, L, shape, 0.25, iters, 5, 0.45, 2,2,2,2,2,2))
print("---- direct pk42-like: d pk42 + J pk42: not modelled in direct probe; dc variants with M pk42:")
rep("dc M pk42, r32 d32, d0 f64, A pk42", run_dc(A, wt, L, shape, 0.25, iters, 5, 0.45, 2,1,1,0,2,1))
rep("dc M32, r pk42, d32", run_dc(A, wt, L, shape, 0.25, iters, 5, 0.45, 1,2,1,0,2,1))
rep("dc M32, r32, d pk42", run_dc(A, wt, L, shape, 0.25, iters, 5, 0.45, 1,1,2,0,2,1))
rep("dc M pk42, r32, d pk42", run_dc(A, wt, L, shape, 0.25, iters, 5, 0.45, 2,1,2,0,2,1))
r1 = run_dc(A, wt, L, shape, 0.25, iters, 5, 0.45, 1,1,1,0,2,1)
e = np.linalg.norm(r1-ref, axis=-1)
print("quantiles of dc error:", np.quantile(e, [0.5, 0.9, 0.99, 0.999, 0.9999]), "sum of top 0.1%:", np.sort(e.ravel())[-e.size//1000:].sum()/e.sum())
