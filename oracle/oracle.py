"""ctypes front-end of the CPU oracle (oracle/fr3d_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py -- never by the product package ``flowreg3d_amd``.

The Python signatures mirror the reference functions they restate
(/root/reference/src/flowreg3d/core/optical_flow_3d.py:22,92,262,319 and
util/resize_util_3D.py:114) so parity tests read like calls into the reference.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FR3D_ORACLE_SO: an alternative build of the same source (e.g. with FMA contraction) for reproducibility studies
_SO = os.environ.get("FR3D_ORACLE_SO") or os.path.join(_HERE, "_build", "libfr3d_oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "fr3d_oracle.c")
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def use_build(name: str = ""):
    """Select the oracle build for this process: "" = default (C library pow), "ppow" = portable pow in the psi
    nonlinearities (bit-comparable with the engine's verification mode), "fma" = FMA-contracted timing build."""
    global _SO, _lib
    _SO = os.path.join(_HERE, "_build", f"libfr3d_oracle{'_' + name if name else ''}.so")
    _lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.fr3d_oracle_resize_tables.restype = C.c_int
        _lib.fr3d_oracle_resize_tables.argtypes = [C.c_int, C.c_int, C.c_double, _ip, _fp]
        _lib.fr3d_oracle_warping_depth.restype = C.c_int
        _lib.fr3d_oracle_warping_depth.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.fr3d_oracle_schedule.restype = C.c_int
        _lib.fr3d_oracle_get_displacement.restype = C.c_int
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _f(a):
    return a.ctypes.data_as(_fp)


def resize_tables(in_len: int, out_len: int, sigma: float):
    """util/resize_util_3D.py:98 _precompute_fused_gauss_cubic"""
    L = lib()
    P = L.fr3d_oracle_resize_tables(in_len, out_len, float(sigma), None, None)
    idx = np.empty((out_len, P), np.int32)
    wt = np.empty((out_len, P), np.float32)
    L.fr3d_oracle_resize_tables(in_len, out_len, float(sigma), idx.ctypes.data_as(_ip), _f(wt))
    return idx, wt


def imresize_fused_gauss_cubic3D(img, size, sigma_coeff=0.6, per_axis=False):
    """util/resize_util_3D.py:114-156 (float and integer images)."""
    img = np.asarray(img)
    od, oh, ow = (int(s) for s in size[:3])
    x = np.ascontiguousarray(img, dtype=np.float32)
    if x.ndim == 3:
        x = x[..., None]
        squeeze = True
    elif x.ndim == 4:
        squeeze = False
    else:
        raise ValueError("img must be 3D or 4D with channels-last")
    D, H, W, Cn = x.shape
    out = np.empty((od, oh, ow, Cn), np.float32)
    for c in range(Cn):
        src = np.ascontiguousarray(x[..., c])
        dst = np.empty((od, oh, ow), np.float32)
        lib().fr3d_oracle_resize3d_ex(_f(src), D, H, W, od, oh, ow, C.c_double(sigma_coeff), int(bool(per_axis)), _f(dst))
        out[..., c] = dst
    if squeeze:
        out = out[..., 0]
    if np.issubdtype(img.dtype, np.integer):  # :150-154 round, clip to the dtype's range, cast
        info = np.iinfo(img.dtype)
        out = np.rint(out)
        np.clip(out, info.min, info.max, out=out)
        return out.astype(img.dtype)
    return out.astype(img.dtype, copy=False)


def spline_filter3(a):
    c = np.array(a, dtype=np.float64, order="C", copy=True)
    lib().fr3d_oracle_spline_filter3(_d(c), *map(C.c_int, c.shape))
    return c


def imregister_wrapper(f2_level, u, v, w, f1_level, interpolation_method="cubic"):
    """core/optical_flow_3d.py:22"""
    f2 = np.asarray(f2_level)
    f1 = np.asarray(f1_level)
    squeeze = f2.ndim == 3
    if squeeze:
        f2 = f2[..., None]
        f1 = f1[..., None]
    m = interpolation_method.lower()
    if m == "cubic":
        order = 3
    elif m == "linear":
        order = 1
    else:
        raise ValueError("Unsupported interpolation method. Use 'linear' or 'cubic'.")
    Z, Y, X, Cn = f2.shape
    f2d = np.ascontiguousarray(f2, dtype=np.float64)
    f1d = np.ascontiguousarray(f1, dtype=np.float64)
    shp = (Z, Y, X)
    ud = np.ascontiguousarray(np.broadcast_to(np.asarray(u, dtype=np.float64), shp))
    vd = np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float64), shp))
    wd = np.ascontiguousarray(np.broadcast_to(np.asarray(w, dtype=np.float64), shp))
    out = np.empty((Z, Y, X, Cn), np.float32)
    lib().fr3d_oracle_imregister(_d(f2d), _d(ud), _d(vd), _d(wd), _d(f1d), Z, Y, X, Cn, order, _f(out))
    return out[..., 0] if Cn == 1 else out


_RAW_CODES = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.uint8): 2, np.dtype(np.uint16): 3,
              np.dtype(np.int16): 4}


def register_raw(vol, flow, ref, interpolation_method="cubic"):
    """registered[t] of the executors (parallelization/sequential_3d.py:153-170): imregister_wrapper on the RAW
    volume `vol` (Z,Y,X,C) in its own dtype with the float32 flow, stored into an array of vol.dtype."""
    vol = np.asarray(vol)
    code = _RAW_CODES[vol.dtype]
    order = 3 if interpolation_method.lower() == "cubic" else 1
    Z, Y, X, Cn = vol.shape
    f2d = np.ascontiguousarray(vol, dtype=np.float64)
    f1d = np.ascontiguousarray(np.asarray(ref).reshape(Z, Y, X, Cn), dtype=np.float64)
    fl = np.asarray(flow, dtype=np.float32)
    ud, vd, wd = (np.ascontiguousarray(fl[..., d], dtype=np.float64) for d in range(3))
    out = np.empty((Z, Y, X, Cn), vol.dtype)
    lib().fr3d_oracle_imregister_typed(_d(f2d), _d(ud), _d(vd), _d(wd), _d(f1d), Z, Y, X, Cn, order, code,
                                       out.ctypes.data_as(C.c_void_p))
    return out


def update_reference(batch_proc, w, reference_proc, interpolation_method="cubic"):
    """BatchMotionCorrector._update_reference (motion_correction/compensate_recording_3D.py:395-429) -> new
    reference_proc (Z,Y,X,C) float64."""
    bp = np.ascontiguousarray(batch_proc, dtype=np.float64)
    fl = np.ascontiguousarray(w, dtype=np.float32)
    rp = np.ascontiguousarray(reference_proc, dtype=np.float64)
    T, Z, Y, X, Cn = bp.shape
    out = np.array(rp, copy=True)
    order = 3 if interpolation_method.lower() == "cubic" else 1
    lib().fr3d_oracle_update_reference(_d(bp), _f(fl), _d(rp), T, Z, Y, X, Cn, order, _d(out))
    return out


def get_motion_tensor_gc(f1, f2, hz, hy, hx):
    """core/optical_flow_3d.py:92 -> (J11,J22,J33,J44,J12,J13,J23,J14,J24,J34)"""
    f1d = np.ascontiguousarray(f1, dtype=np.float64)
    f2d = np.ascontiguousarray(f2, dtype=np.float64)
    Z, Y, X = f1d.shape
    Js = [np.empty((Z + 2, Y + 2, X + 2), np.float64) for _ in range(10)]
    arr = (_dp * 10)(*[_d(j) for j in Js])
    lib().fr3d_oracle_motion_tensor_gc(_d(f1d), _d(f2d), Z, Y, X, C.c_double(hz), C.c_double(hy),
                                       C.c_double(hx), arr)
    return tuple(Js)


def compute_flow_3d(J11, J22, J33, J44, J12, J13, J23, J14, J24, J34, weight, u, v, w, alpha_x,
                    alpha_y, alpha_z, iterations, update_lag, a_data, a_smooth, hx, hy, hz):
    """core/level_solver_3d.py:314 -> (P,M,N,3)"""
    Js = [np.ascontiguousarray(j, dtype=np.float64) for j in
          (J11, J22, J33, J44, J12, J13, J23, J14, J24, J34)]
    P, M, N, Cn = Js[0].shape
    wt = np.ascontiguousarray(weight, dtype=np.float64)
    ud, vd, wd = (np.ascontiguousarray(a, dtype=np.float64) for a in (u, v, w))
    ad = np.ascontiguousarray(np.broadcast_to(np.asarray(a_data, dtype=np.float64), (Cn,)))
    out = np.empty((P, M, N, 3), np.float64)
    arr = (_dp * 10)(*[_d(j) for j in Js])
    lib().fr3d_oracle_compute_flow_3d(arr, _d(wt), _d(ud), _d(vd), _d(wd), P, M, N, Cn,
                                      C.c_double(alpha_x), C.c_double(alpha_y), C.c_double(alpha_z),
                                      int(iterations), int(update_lag), _d(ad), C.c_double(a_smooth),
                                      C.c_double(hx), C.c_double(hy), C.c_double(hz), _d(out))
    return out


def level_solver(J11, J22, J33, J44, J12, J13, J23, J14, J24, J34, weight, u, v, w, alpha,
                 iterations, update_lag, verbose, a_data, a_smooth, hx, hy, hz):
    """core/optical_flow_3d.py:262"""
    r = compute_flow_3d(J11, J22, J33, J44, J12, J13, J23, J14, J24, J34, weight, u, v, w,
                        alpha[0], alpha[1], alpha[2], iterations, update_lag, a_data, a_smooth,
                        hx, hy, hz)
    return r[..., 0], r[..., 1], r[..., 2]


def median5(a):
    """scipy.ndimage.median_filter(a, size=(5,5,5), mode='mirror')"""
    ad = np.ascontiguousarray(a, dtype=np.float64)
    out = np.empty_like(ad)
    lib().fr3d_oracle_median5(_d(ad), *map(C.c_int, ad.shape), _d(out))
    return out


def warpingDepth(eta, levels, p, m, n):
    """core/optical_flow_3d.py:77"""
    return lib().fr3d_oracle_warping_depth(float(eta), int(levels), int(p), int(m), int(n))


def schedule(p, m, n, eta, levels, min_level):
    """Level sizes (coarse -> fine) and the effective min_level (core/optical_flow_3d.py:389-408)."""
    sizes = np.zeros((256, 3), np.int32)
    eff = C.c_int(0)
    cnt = lib().fr3d_oracle_schedule(int(p), int(m), int(n), C.c_double(eta), int(levels),
                                     int(min_level), sizes.ctypes.data_as(_ip), 256, C.byref(eff))
    return [tuple(int(v) for v in sizes[i]) for i in range(cnt)], eff.value


def expand_weight(weight, p, m, n, n_channels):
    """core/optical_flow_3d.py:351-381"""
    if weight is None:
        return np.ones((p, m, n, n_channels), np.float64) / n_channels
    weight = np.asarray(weight).astype(np.float64)
    if weight.ndim < 4:
        if weight.ndim == 1:
            if len(weight) < n_channels:
                we = np.full(n_channels, 1.0 / n_channels, np.float64)
                we[: len(weight)] = weight
                weight = we
            elif len(weight) > n_channels:
                weight = weight[:n_channels]
            weight = weight / weight.sum()
            weight = np.ones((p, m, n, n_channels), np.float64) * weight.reshape(1, 1, 1, -1)
        else:
            weight = np.ones((p, m, n, n_channels), np.float64) * weight[..., np.newaxis]
    return weight


def get_displacement(fixed, moving, alpha=(2, 2, 2), update_lag=10, iterations=20, min_level=0,
                     levels=50, eta=0.8, a_smooth=0.5, a_data=0.45, const_assumption="gc",
                     uvw=None, weight=None):
    """core/optical_flow_3d.py:319 -> (Z,Y,X,3) float64, components [dx,dy,dz]"""
    fixed = np.asarray(fixed).astype(np.float64)
    moving = np.asarray(moving).astype(np.float64)
    if fixed.ndim == 3:
        fixed = fixed[..., None]
        moving = moving[..., None]
    p, m, n, Cn = fixed.shape
    fixed = np.ascontiguousarray(fixed)
    moving = np.ascontiguousarray(moving)
    wt = np.ascontiguousarray(expand_weight(weight, p, m, n, Cn))
    ad = np.ascontiguousarray(np.broadcast_to(np.asarray(a_data, dtype=np.float64), (Cn,)))
    al = np.ascontiguousarray(np.asarray(alpha, dtype=np.float64).reshape(3))
    uv = None if uvw is None else np.ascontiguousarray(uvw, dtype=np.float64)
    flow = np.empty((p, m, n, 3), np.float64)
    rc = lib().fr3d_oracle_get_displacement(
        _d(fixed), _d(moving), p, m, n, Cn, _d(al), int(update_lag), int(iterations),
        int(min_level), int(levels), C.c_double(eta), C.c_double(a_smooth), _d(ad),
        None if uv is None else _d(uv), _d(wt), _d(flow))
    if rc != 0:
        raise ValueError("fr3d_oracle_get_displacement: bad arguments")
    return flow


# ---- f-1 preprocessing (util/image_processing_3D.py) ---------------------------------------------

def gaussian_filter3(vol, sigma_zyx, truncate=4.0):
    """scipy.ndimage.gaussian_filter(vol, sigma=(sz,sy,sx), mode="reflect", truncate=truncate), fp64."""
    a = np.array(vol, dtype=np.float64, order="C", copy=True)
    s = (C.c_double * 3)(*[float(x) for x in sigma_zyx])
    lib().fr3d_oracle_gaussian_filter3(_d(a), *map(C.c_int, a.shape), s, C.c_double(truncate))
    return a


def normalize(arr, ref=None, channel_normalization="together", eps=1e-8):
    """util/image_processing_3D.py:12-92 (host arithmetic only; restated for the parity tests)."""
    arr = np.asarray(arr)
    if channel_normalization == "separate" and arr.ndim in (4, 5):
        result = np.zeros_like(arr, dtype=np.float64)
        for c in range(arr.shape[-1]):
            src = ref[..., c] if (ref is not None and ref.ndim >= 4) else arr[..., c]
            lo, hi = src.min(), src.max()
            rng = hi - lo
            result[..., c] = (arr[..., c] - lo) / rng if rng > 0 else arr[..., c] - lo
        return result
    src = ref if ref is not None else arr
    lo, hi = src.min(), src.max()
    if channel_normalization == "separate":
        rng = hi - lo
        return (arr - lo) / rng if rng > 0 else arr - lo
    return (arr - lo) / (hi - lo + eps)


def apply_gaussian_filter(arr, sigma, mode="reflect", truncate=4.0):
    """util/image_processing_3D.py:95-162 for (Z,Y,X,C) and (T,Z,Y,X,C) with sigma (4,) or (C,4) =
    [sx,sy,sz,st]; the temporal axis is filtered by the same symmetric correlate."""
    assert mode == "reflect"
    arr = np.asarray(arr)
    sigma = np.asarray(sigma, dtype=np.float64)
    out = np.zeros(arr.shape, np.float64)
    nc = arr.shape[-1]
    for c in range(nc):
        s = sigma[min(c, len(sigma) - 1)] if sigma.ndim == 2 else sigma
        if arr.ndim == 4:
            out[..., c] = gaussian_filter3(arr[..., c], (s[2], s[1], s[0]), truncate)
        else:
            T = arr.shape[0]
            vols = np.stack([arr[t, ..., c].astype(np.float64) for t in range(T)])
            st = s[3] if len(s) == 4 else 0.0
            if st > 1e-15:  # axis 0 of the 4-D filter comes first
                radius = lib().fr3d_oracle_gaussian_kernel(C.c_double(st), C.c_double(truncate), None, 0)
                w = np.empty(2 * radius + 1, np.float64)
                lib().fr3d_oracle_gaussian_kernel(C.c_double(st), C.c_double(truncate), _d(w), w.size)
                flat = np.ascontiguousarray(vols.reshape(T, 1, -1))
                res = np.empty_like(flat)
                lib().fr3d_oracle_correlate1d_sym(_d(flat), T, 1, flat.shape[2], 0, _d(w), radius, _d(res))
                vols = res.reshape(vols.shape)
            for t in range(T):
                out[t, ..., c] = gaussian_filter3(vols[t], (s[2], s[1], s[0]), truncate)
    return out
