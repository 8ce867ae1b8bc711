"""Host-side mirror of flowreg3d's flow-engine functions over the HIP C ABI.

Same names, argument meaning, defaults, return shapes/dtypes and error behaviour as
``flowreg3d.core.optical_flow_3d`` (get_displacement :319, imregister_wrapper :22, level_solver
:262, get_motion_tensor_gc :92, warpingDepth :77, add_boundary :88) and
``flowreg3d.util.resize_util_3D.imresize_fused_gauss_cubic3D`` (:114), so they can be passed as
``get_displacement_func`` / ``imregister_func`` to the reference's executors or monkey-patched in.
All arithmetic runs in hand-written gfx950 kernels (flowreg3d_amd/csrc); this file only reshapes,
casts and validates.  No CPU fallback exists: without the built library / a GPU these raise.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

__all__ = ["get_displacement", "imregister_wrapper", "level_solver", "get_motion_tensor_gc",
           "imresize_fused_gauss_cubic3D", "tensor_factors", "warpingDepth", "add_boundary", "pyramid_schedule",
           "expand_weight", "median_filter5"]


def add_boundary(f):
    """core/optical_flow_3d.py:88"""
    return np.pad(f, 1, mode="edge")


def warpingDepth(eta, levels, p, m, n):
    """core/optical_flow_3d.py:77-85 (pure host logic, no GPU needed)."""
    min_dim = min(p, m, n)
    warpingdepth = 0
    for _ in range(levels):
        warpingdepth += 1
        min_dim *= eta
        if round(min_dim) < 10:
            break
    return warpingdepth


def pyramid_schedule(p, m, n, eta, levels, min_level=0):
    """Level sizes (coarse -> fine) and the effective min_level, computed by the engine's own
    host code (fr3d_schedule; core/optical_flow_3d.py:389-408)."""
    lib = _lib.load()
    sizes = np.zeros((256, 3), np.int32)
    eff = C.c_int(0)
    cnt = lib.fr3d_schedule(int(p), int(m), int(n), float(eta), int(levels), int(min_level),
                            sizes.ctypes.data_as(C.POINTER(C.c_int)), 256, C.byref(eff))
    if cnt < 0:
        raise ValueError(_lib.last_error())
    return [tuple(int(v) for v in sizes[i]) for i in range(cnt)], int(eff.value)


def expand_weight(weight, p, m, n, n_channels):
    """Weight handling of get_displacement (core/optical_flow_3d.py:351-381) -> (p,m,n,C) float64,
    or None for the default 1/C (the engine then builds the constant itself)."""
    if weight is None:
        return None
    weight = np.asarray(weight).astype(np.float64)
    if weight.ndim < 4:
        if weight.ndim == 1:
            if len(weight) < n_channels:
                expanded = np.full(n_channels, 1.0 / n_channels, dtype=np.float64)
                expanded[: len(weight)] = weight
                weight = expanded
            elif len(weight) > n_channels:
                weight = weight[:n_channels]
            weight = weight / weight.sum()
            weight = np.ones((p, m, n, n_channels), dtype=np.float64) * weight.reshape(1, 1, 1, -1)
        else:
            weight = np.ones((p, m, n, n_channels), dtype=np.float64) * weight[..., np.newaxis]
    if weight.shape != (p, m, n, n_channels):
        raise ValueError(f"weight has shape {weight.shape}, expected {(p, m, n, n_channels)}")
    return weight


def is_default_weight(weight, n_channels):
    """True for weights that expand to the constant 1/C the reference builds for weight=None (core/optical_flow_3d.py:351-381:
    a 1-D weight is completed with 1/C, cut to C entries and normalised): callers then pass NULL and the engine makes the
    constant itself instead of receiving a (Z,Y,X,C) array of it."""
    if weight is None:
        return True
    w = np.asarray(weight, dtype=np.float64)
    if w.ndim != 1 or w.size == 0:
        return False
    if len(w) < n_channels:
        w = np.concatenate([w, np.full(n_channels - len(w), 1.0 / n_channels)])
    w = w[:n_channels]
    return bool(w.sum() != 0 and np.all(w / w.sum() == 1.0 / n_channels))


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def get_displacement(fixed, moving, alpha=(2, 2, 2), update_lag=10, iterations=20, min_level=0,
                     levels=50, eta=0.8, a_smooth=0.5, a_data=0.45, const_assumption="gc",
                     uvw=None, weight=None, solver_fp64=None, solver_sweep=0):
    """core/optical_flow_3d.py:319-542 -> (Z,Y,X,3) float64 with components [dx,dy,dz].

    fixed/moving are used by the reference only through the fp32 resampler
    (util/resize_util_3D.py:116), and uvw/weight likewise, so they cross the ABI as float32.
    ``solver_fp64`` is an extension (fr3d_params.solver_fp64): None = automatic (one channel: fp32 solver storage
    with fp64 update arithmetic up to 2^22 voxels, packed 42-bit storage above; several channels: fp64 storage),
    0 / 1 / 2 / 3 force fp32 / fp64 arithmetic on fp32 storage / fp64 storage / packed 42-bit storage.
    ``solver_sweep`` (fr3d_params.solver_sweep): 0 = the engine's choice, 1 = one launch per hyperplane step,
    2 = the window kernel (a psi window per workgroup on chip); the results are bit-identical.
    """
    fixed = np.asarray(fixed)
    moving = np.asarray(moving)
    if fixed.ndim == 3:
        fixed = fixed[..., None]
        moving = moving[..., None]
    if fixed.ndim != 4 or moving.shape != fixed.shape:
        raise ValueError("fixed and moving must have the same (Z,Y,X[,C]) shape")
    p, m, n, nc = fixed.shape
    wt = None if is_default_weight(weight, nc) else expand_weight(weight, p, m, n, nc)
    params = _lib.make_params(alpha, update_lag, iterations, min_level, levels, eta, a_smooth, a_data, nc,
                              solver_fp64, solver_sweep)
    f32, m32 = _f32c(fixed), _f32c(moving)
    u32 = None
    if uvw is not None:
        u32 = _f32c(uvw)
        if u32.shape != (p, m, n, 3):
            raise ValueError(f"uvw has shape {u32.shape}, expected {(p, m, n, 3)}")
    w32 = None if wt is None else _f32c(wt)
    flow = np.empty((p, m, n, 3), np.float32)
    lib = _lib.init()
    _lib.check(lib.fr3d_get_displacement(C.byref(params), _lib.ptr(f32), _lib.ptr(m32), p, m, n, nc,
                                         _lib.ptr(u32), _lib.ptr(w32), _lib.ptr(flow)))
    _lib.warn_if_degraded()
    return flow.astype(np.float64)


def _order_of(interpolation_method):
    m = str(interpolation_method).lower()
    if m == "cubic":
        return 3
    if m == "linear":
        return 1
    raise ValueError("Unsupported interpolation method. Use 'linear' or 'cubic'.")


def imregister_wrapper(f2_level, u, v, w, f1_level, interpolation_method="cubic"):
    """core/optical_flow_3d.py:22-74 -> float32, channel axis dropped when C == 1."""
    order = _order_of(interpolation_method)
    f2 = np.asarray(f2_level)
    f1 = np.asarray(f1_level)
    if f2.ndim == 3:
        f2 = f2[..., None]
        f1 = f1[..., None]
    Z, Y, X, nc = f2.shape
    # volumes: float64 only when float32 would change values (the reference filters in float64)
    if f2.dtype == np.float64 or f1.dtype == np.float64:
        vdt, vt = _lib.F64, np.float64
    else:
        vdt, vt = _lib.F32, np.float32
    vol = np.ascontiguousarray(f2, dtype=vt)
    ref = np.ascontiguousarray(f1, dtype=vt)
    comps = [np.broadcast_to(np.asarray(a), (Z, Y, X)) for a in (u, v, w)]
    if any(c.dtype == np.float64 for c in comps):
        fdt, ft = _lib.F64, np.float64
    else:
        fdt, ft = _lib.F32, np.float32
    flow = np.empty((Z, Y, X, 3), ft)
    for d in range(3):
        flow[..., d] = comps[d]
    out = np.empty((Z, Y, X, nc), np.float32)
    lib = _lib.init()
    _lib.check(lib.fr3d_warp(_lib.ptr(vol), vdt, _lib.ptr(flow), fdt, _lib.ptr(ref), Z, Y, X, nc, order,
                             _lib.ptr(out)))
    if nc == 1:
        out = out[..., 0]
    return out


def imresize_fused_gauss_cubic3D(img, size, sigma_coeff=0.6, per_axis=False):
    """util/resize_util_3D.py:114-156: fused Gauss x cubic separable resampling on the device.  Integer
    images are rounded and clipped to their dtype's range on the way back, like the reference (:150-154)."""
    img = np.asarray(img)
    if img.ndim not in (3, 4):
        raise ValueError("img must be 3D or 4D with channels-last")
    od, oh, ow = (int(s) for s in size[:3])
    x = img.astype(np.float32, copy=False)
    x4 = x[..., None] if x.ndim == 3 else x
    D, H, W, nc = x4.shape
    out = np.empty((od, oh, ow, nc), np.float32)
    lib = _lib.init()
    for c in range(nc):
        src = np.ascontiguousarray(x4[..., c])
        dst = np.empty((od, oh, ow), np.float32)
        _lib.check(lib.fr3d_resize3d_ex(_lib.ptr(src), D, H, W, od, oh, ow, float(sigma_coeff), int(bool(per_axis)),
                                        _lib.ptr(dst)), value_error=True)
        out[..., c] = dst
    if x.ndim == 3:
        out = out[..., 0]
    if np.issubdtype(img.dtype, np.integer):
        info = np.iinfo(img.dtype)
        out = np.rint(out)
        np.clip(out, info.min, info.max, out=out)
        return out.astype(img.dtype)
    return out.astype(img.dtype, copy=False)


def get_displacement_verify(fixed, moving, alpha=(2, 2, 2), update_lag=10, iterations=20, min_level=0,
                            levels=50, eta=0.8, a_smooth=1.0, a_data=0.45, uvw=None, weight=None):
    """Verification mode of ``get_displacement`` (fr3d_get_displacement_verify; both solver branches, a_smooth == 1 and the
    psi_smooth branch of level_solver_3d.py:262-311,400-471): the reference's own
    arithmetic (core/level_solver_3d.py:356-377,472-540: fp64, expanded quadratic form, per-channel order, true
    divisions) on the engine's data path, the level flow kept in float64 like the reference's.  Bit-identical to the
    CPU restatement of the reference built with the same portable pow (its ``ppow`` build); ~5x slower than the fp64-storage mode.
    -> (Z,Y,X,3) float64."""
    fixed = np.asarray(fixed)
    moving = np.asarray(moving)
    if fixed.ndim == 3:
        fixed = fixed[..., None]
        moving = moving[..., None]
    if fixed.ndim != 4 or moving.shape != fixed.shape:
        raise ValueError("fixed and moving must have the same (Z,Y,X[,C]) shape")
    p, m, n, nc = fixed.shape
    wt = None if is_default_weight(weight, nc) else expand_weight(weight, p, m, n, nc)
    params = _lib.make_params(alpha, update_lag, iterations, min_level, levels, eta, a_smooth, a_data, nc, 2)
    f32, m32 = _f32c(fixed), _f32c(moving)
    u32 = None
    if uvw is not None:
        u32 = _f32c(uvw)
        if u32.shape != (p, m, n, 3):
            raise ValueError("uvw must have shape (Z,Y,X,3)")
    w32 = None if wt is None else _f32c(wt)
    out = np.empty((p, m, n, 3), np.float64)
    lib = _lib.init()
    _lib.check(lib.fr3d_get_displacement_verify(C.byref(params), _lib.ptr(f32), _lib.ptr(m32), p, m, n, nc, _lib.ptr(u32),
                                                _lib.ptr(w32), _lib.ptr(out)))
    return out


def get_motion_tensor_gc(f1, f2, hz, hy, hx, return_factors=False):
    """core/optical_flow_3d.py:92-152 -> 10 arrays (Z+2,Y+2,X+2) float64, outer ring zero, in the
    order J11,J22,J33,J44,J12,J13,J23,J14,J24,J34.  Values are the engine's fp32 storage.
    ``return_factors`` additionally returns the (12,Z,Y,X) fp32 square-root factors the solver
    evaluates psi_data from (include/flowreg3d_hip.h, fr3d_motion_tensor)."""
    a, b = _f32c(f1), _f32c(f2)
    if a.ndim != 3 or a.shape != b.shape:
        raise ValueError("f1 and f2 must be 3-D arrays of equal shape")
    Z, Y, X = a.shape
    J = np.empty((10, Z, Y, X), np.float32)
    A = np.empty((12, Z, Y, X), np.float32) if return_factors else None
    lib = _lib.init()
    _lib.check(lib.fr3d_motion_tensor(_lib.ptr(a), _lib.ptr(b), Z, Y, X, float(hz), float(hy), float(hx),
                                      _lib.ptr(J), _lib.ptr(A)))
    out = tuple(np.pad(J[k].astype(np.float64), 1) for k in range(10))
    return (out, A) if return_factors else out


class TensorRankError(ValueError):
    """the motion tensor is not (numerically) of rank <= 3: no square-root factors"""


def tensor_factors(J11, J22, J33, J44, J12, J13, J23, J14, J24, J34):
    """Rank-3 square-root factors A (12, ...) with J = sum_k a_k a_k^T (k = 0..2, a_k in R^4) of the
    symmetric PSD 4x4 motion tensor given entry-wise (any leading shape), in float64.

    The reference's tensor is a sum of three outer products (core/optical_flow_3d.py:134-143), so
    the three leading eigenpairs reproduce it; psi_data's quadratic form d^T J d then becomes
    sum_k (a_k . d)^2, which is how the device evaluates it.  Raises TensorRankError for a tensor of higher
    rank (level_solver then solves on the tensor entries instead)."""
    raw = [np.asarray(j) for j in (J11, J22, J33, J44, J12, J13, J23, J14, J24, J34)]
    Js = [j.astype(np.float64, copy=False) for j in raw]
    shp = Js[0].shape
    # relative size of the smallest eigenvalue a genuine rank-3 tensor may show: entries that are float32, or
    # float64 holding float32-rounded values (get_motion_tensor_gc returns the engine's fp32 storage), carry
    # ~6e-8 of rounding each, which lifts lambda_min to ~1e-7 of the trace; exact float64 tensors stay at ~1e-16
    f32_grade = any(j.dtype != np.float64 or np.array_equal(j, j.astype(np.float32)) for j in raw)
    rank_tol = 32 * float(np.finfo(np.float32).eps) if f32_grade else 1e-9
    M = np.empty(shp + (4, 4), np.float64)
    idx = {(0, 0): 0, (1, 1): 1, (2, 2): 2, (3, 3): 3, (0, 1): 4, (0, 2): 5, (1, 2): 6, (0, 3): 7, (1, 3): 8,
           (2, 3): 9}
    for (r, c), k in idx.items():
        M[..., r, c] = Js[k]
        M[..., c, r] = Js[k]
    lam, V = np.linalg.eigh(M)          # ascending
    # the factor solver keeps three factors: a tensor that is not (numerically) rank <= 3 -- e.g. one built by
    # another constancy assumption -- would silently be solved as a different system
    trace = np.maximum(lam.sum(axis=-1), 0.0)
    if np.any(np.abs(lam[..., 0]) > rank_tol * trace + 1e-300):
        raise TensorRankError(f"the motion tensor is not rank 3 (smallest eigenvalue exceeds {rank_tol:.1e} of the trace)")
    lam = np.clip(lam[..., 1:], 0.0, None)  # three leading eigenvalues
    V = V[..., :, 1:]
    A = np.sqrt(lam)[..., None, :] * V  # (..., 4, 3): column k = a_k
    A = np.moveaxis(A, (-1, -2), (0, 1)).reshape((12,) + shp)  # index 4k + column
    return A


def level_solver(J11, J22, J33, J44, J12, J13, J23, J14, J24, J34, weight, u, v, w, alpha, iterations,
                 update_lag, verbose, a_data, a_smooth, hx, hy, hz, solver_fp64=False):
    """core/optical_flow_3d.py:262-316 -> (du, dv, dw), each (P,M,N) float64.

    J*, weight: (P,M,N,C) with the zero outer ring; u,v,w: (P,M,N) edge-padded level flow.  Only the
    interior is solved (as in the reference); the returned ghost ring is the edge pad of the
    interior (the reference leaves the Neumann copy of the previous iterate there; it is never read
    downstream, core/optical_flow_3d.py:517-535).

    The rank-3 gradient-constancy tensor of get_motion_tensor_gc takes the production solver (square-root factors,
    fr3d_level_solve).  ANY OTHER tensor -- the reference accepts whatever its caller built -- and u, v, w whose ghost
    ring is not the edge pad of the interior are solved on the entries by fr3d_level_solve_tensor: the reference's own
    arithmetic in fp64 (the verification mode's sweep; slower)."""
    Js = [np.asarray(j) for j in (J11, J22, J33, J44, J12, J13, J23, J14, J24, J34)]
    if Js[0].ndim == 3:
        Js = [j[..., None] for j in Js]
    P, M, N, nc = Js[0].shape
    wt = np.asarray(weight)
    if wt.ndim == 3:
        wt = wt[..., None]
    inner = (slice(1, -1),) * 3
    edge_padded = True
    for name, a in (("u", u), ("v", v), ("w", w)):
        a = np.asarray(a)
        if a.shape != (P, M, N):
            raise ValueError(f"{name} must have shape {(P, M, N)}")
        # the production kernels take the ghost ring of u as the edge pad of its interior (add_boundary, :88); a
        # caller-chosen ring enters the surface voxels' stencil and psi_smooth: solved on the entries path below
        edge_padded = edge_padded and np.array_equal(a, np.pad(a[inner], 1, mode="edge"))
    Ji = [np.moveaxis(j[inner], -1, 0) for j in Js]  # each (C,Z,Y,X)
    wd = _f32c(np.moveaxis(wt[inner], -1, 0))
    al = (C.c_double * 3)(*[float(x) for x in alpha])
    ad_np = np.broadcast_to(np.asarray(a_data, dtype=np.float64).reshape(-1), (nc,))
    ad = (C.c_double * nc)(*[float(x) for x in ad_np])
    lib = _lib.init()
    try:
        if not edge_padded:
            raise TensorRankError("ghost ring of u, v, w is not the edge pad")
        Ad = _f32c(tensor_factors(*Ji))  # (12,C,Z,Y,X)
    except TensorRankError:
        Jin = np.ascontiguousarray(np.stack([np.stack([Ji[q][c] for q in range(10)]) for c in range(nc)]), np.float64)
        uvw64 = np.ascontiguousarray(np.stack([np.asarray(a)[inner] for a in (u, v, w)], 0), np.float64)
        ring = None if edge_padded else np.ascontiguousarray(np.stack([np.asarray(a) for a in (u, v, w)], 0), np.float64)
        out64 = np.empty((3, P - 2, M - 2, N - 2), np.float64)
        _lib.check(lib.fr3d_level_solve_tensor(_lib.ptr(Jin), _lib.ptr(wd), _lib.ptr(uvw64),
                                               None if ring is None else _lib.ptr(ring), P - 2, M - 2, N - 2, nc, al,
                                               int(iterations), int(update_lag), ad, float(a_smooth), float(hx), float(hy),
                                               float(hz), _lib.ptr(out64)))
        return tuple(np.pad(out64[d], 1, mode="edge") for d in range(3))
    uvw = np.stack([np.asarray(a)[inner] for a in (u, v, w)], 0).astype(np.float32)
    out = np.empty((3, P - 2, M - 2, N - 2), np.float32)
    _lib.check(lib.fr3d_level_solve(_lib.ptr(Ad), _lib.ptr(wd), _lib.ptr(uvw), P - 2, M - 2, N - 2,
                                    nc, al, int(iterations), int(update_lag), ad, float(a_smooth), float(hx),
                                    float(hy), float(hz), 1 if solver_fp64 else 0, _lib.ptr(out)))
    return tuple(np.pad(out[d].astype(np.float64), 1, mode="edge") for d in range(3))


def median_filter5(a):
    """scipy.ndimage.median_filter(a, size=(5,5,5), mode='mirror') on the device (fp32)."""
    x = _f32c(a)
    if x.ndim != 3:
        raise ValueError("3-D array expected")
    out = np.empty_like(x)
    lib = _lib.init()
    _lib.check(lib.fr3d_median5(_lib.ptr(x), *x.shape, _lib.ptr(out)))
    return out
