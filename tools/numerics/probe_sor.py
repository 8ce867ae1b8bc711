"""Explore the SOR kernel's number formats on the CPU (development tool)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as o
from flowreg3d_amd.synthetic import make_pair

lib = C.CDLL("/tmp/probe_sor.so")
dp = C.POINTER(C.c_double)

def factors(f1, f2, hz, hy, hx):
    f1p = np.pad(f1, 1, mode="symmetric"); f2p = np.pad(f2, 1, mode="symmetric")
    gz1, gy1, gx1 = np.gradient(f1p, hz, hy, hx); gz2, gy2, gx2 = np.gradient(f2p, hz, hy, hx)
    rep = lambda a: np.pad(a[1:-1, 1:-1, 1:-1], 1, mode="symmetric")
    fx, fy, fz, ft = rep(0.5*(gx1+gx2)), rep(0.5*(gy1+gy2)), rep(0.5*(gz1+gz2)), rep(f2p-f1p)
    dfx = np.gradient(fx, hz, hy, hx); dfy = np.gradient(fy, hz, hy, hx); dft = np.gradient(ft, hz, hy, hx)
    fxy, fxz, fyz = dfx[1], dfx[0], dfy[0]; fzt, fyt, fxt = dft
    def g3(f):
        a = np.zeros_like(f); b = np.zeros_like(f); c = np.zeros_like(f)
        a[:, :, 1:-1] = (f[:, :, :-2] - 2*f[:, :, 1:-1] + f[:, :, 2:])/hx**2
        b[:, 1:-1] = (f[:, :-2] - 2*f[:, 1:-1] + f[:, 2:])/hy**2
        c[1:-1] = (f[:-2] - 2*f[1:-1] + f[2:])/hz**2
        return a, b, c
    a1, b1, c1 = g3(f1p); a2, b2, c2 = g3(f2p)
    fxx, fyy, fzz = 0.5*(a1+a2), 0.5*(b1+b2), 0.5*(c1+c2)
    rx = 1/((np.sqrt(fxx**2+fxy**2+fxz**2)**2)+1e-6); ry = 1/((np.sqrt(fxy**2+fyy**2+fyz**2)**2)+1e-6)
    rz = 1/((np.sqrt(fxz**2+fyz**2+fzz**2)**2)+1e-6)
    sx, sy, sz = np.sqrt(rx), np.sqrt(ry), np.sqrt(rz)
    A = [sx*fxx, sx*fxy, sx*fxz, sx*fxt, sy*fxy, sy*fyy, sy*fyz, sy*fyt, sz*fxz, sz*fyz, sz*fzz, sz*fzt]
    return [np.ascontiguousarray(a[1:-1, 1:-1, 1:-1]) for a in A]

def run(J, A, wt, L, dims, ax, iters, lag, adc, flags):
    Z, Y, X = dims
    out = [np.zeros(dims) for _ in range(3)]
    arr = lambda lst: (dp*len(lst))(*[a.ctypes.data_as(dp) for a in lst])
    lib.sor_probe(arr(J), arr(A), wt.ctypes.data_as(dp), arr(L), Z, Y, X, C.c_double(ax), C.c_double(ax), C.c_double(ax),
                  iters, lag, C.c_double(adc), flags, arr(out))
    return np.stack(out, -1)

if __name__ == "__main__":
    shape = (32, 64, 64)
    f1, f2, gt = make_pair(shape, seed=1234, scale=0.3)
    f1 = f1.astype(np.float64); f2 = f2.astype(np.float64)
    Jp = o.get_motion_tensor_gc(f1, f2, 1.0, 1.0, 1.0)
    J = [np.ascontiguousarray(j[1:-1, 1:-1, 1:-1]) for j in Jp]
    A = factors(f1, f2, 1.0, 1.0, 1.0)
    # factor consistency
    print("J11 from factors maxdiff", np.abs(A[0]**2+A[4]**2+A[8]**2 - J[0]).max(), "J14", np.abs(A[0]*A[3]+A[4]*A[7]+A[8]*A[11]-J[7]).max())
    wt = np.ones(shape); L = [np.zeros(shape) for _ in range(3)]
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    ref = run(J, A, wt, L, shape, 0.25, iters, 5, 0.45, 0)
    # cross-check the probe against the oracle solver
    z = np.zeros((34, 66, 66))
    orc = o.compute_flow_3d(*[j[..., None] for j in Jp], np.pad(wt, 1)[..., None], z, z, z, 0.25, 0.25, 0.25, iters, 5, np.array([0.45]), 1.0, 1, 1, 1)[1:-1, 1:-1, 1:-1]
    print("probe(all fp64) vs oracle: max", np.abs(ref-orc).max(), " |d| max", np.abs(ref).max())
    names = {1: "J32", 2: "d32", 4: "arith32(+d32)", 8: "psi from factors", 16: "L32", 32: "wpsi32", 64: "J from factors"}
    for flags in (1, 2, 32, 1|2|32, 1|2|4|32, 8, 1|8, 1|8|2|32, 1|8|4|32, 1|8|64, 1|8|64|2|32, 1|8|64|4|32):
        r = run(J, A, wt, L, shape, 0.25, iters, 5, 0.45, flags)
        e = np.linalg.norm(r-ref, axis=-1)
        desc = "+".join(v for k, v in names.items() if flags & k)
        print(f"flags={flags:3d} {desc:55s} EPE mean {e.mean():.3e} max {e.max():.3e}")
