"""Device mirror of flowreg3d's pre-flow preprocessing.

Reference: ``util/image_processing_3D.py`` ``normalize`` (:12-92) and ``apply_gaussian_filter``
(:95-162), composed as ``BatchMotionCorrector._preprocess_frames`` does
(``motion_correction/compensate_recording_3D.py:229-254``: normalize -> filter, float64 out).
The min/max search of ``normalize`` stays a NumPy reduction on the host (it runs on the reference
volume, once); the per-voxel work -- affine map and the separable Gaussian -- runs in HIP
(``fr3d_preprocess``).  Same names and argument meaning as the reference.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

_DTYPES = {np.dtype(np.float32): _lib.F32, np.dtype(np.float64): _lib.F64, np.dtype(np.uint8): _lib.U8,
           np.dtype(np.uint16): _lib.U16, np.dtype(np.int16): _lib.I16}


def _norm_constants(arr, ref, channel_normalization, eps):
    """(min, den) per channel so that normalize(arr) == (arr - min) / den exactly as :12-92 computes it."""
    nc = arr.shape[-1]
    if channel_normalization == "separate":
        mins, dens = [], []
        for c in range(nc):
            src = ref[..., c] if (ref is not None and ref.ndim >= 4) else arr[..., c]
            lo, hi = float(src.min()), float(src.max())
            rng = hi - lo
            mins.append(lo)
            dens.append(rng if rng > 0 else 1.0)
        return np.array(mins), np.array(dens)
    src = ref if ref is not None else arr
    lo, hi = float(src.min()), float(src.max())
    return np.full(nc, lo), np.full(nc, hi - lo + eps)


def _sigma_table(sigma, nc):
    sigma = np.asarray(sigma, dtype=np.float64)
    tab = np.zeros((nc, 4), np.float64)
    for c in range(nc):
        s = sigma[min(c, len(sigma) - 1)] if sigma.ndim == 2 else sigma
        s = np.asarray(s, dtype=np.float64).reshape(-1)
        if s.size == 1:
            s = np.repeat(s, 3)
        tab[c, : min(4, s.size)] = s[:4]
    return tab


# scipy.ndimage boundary modes -> FR3D_BOUNDARY_* (include/flowreg3d_hip.h)
_MODES = {"reflect": 0, "grid-mirror": 0, "constant": 1, "grid-constant": 1, "nearest": 2, "mirror": 3, "wrap": 4,
          "grid-wrap": 4}


def _run(arr, mins, dens, sigma, truncate, spatial_only, out_dtype, mode="reflect"):
    a = np.asarray(arr)
    squeeze_t = a.ndim == 4
    if a.ndim not in (4, 5):
        raise ValueError("expected (Z,Y,X,C) or (T,Z,Y,X,C)")
    if squeeze_t:
        a = a[None]
    if a.dtype not in _DTYPES:
        a = a.astype(np.float64)
    a = np.ascontiguousarray(a)
    T, Z, Y, X, nc = a.shape
    tab = _sigma_table(sigma, nc)
    if spatial_only or squeeze_t:
        tab[:, 3] = 0.0  # (Z,Y,X,C) input: 3-D filter only (:113-133)
    out = np.empty(a.shape, np.float32 if out_dtype == _lib.F32 else np.float64)
    lib = _lib.init()
    dp = C.POINTER(C.c_double)
    mins = np.ascontiguousarray(mins, dtype=np.float64)
    dens = np.ascontiguousarray(dens, dtype=np.float64)
    if mode not in _MODES:
        raise RuntimeError(f"boundary mode not supported: {mode!r}")  # scipy raises RuntimeError for an unknown mode
    _lib.check(lib.fr3d_gaussian_filter(_lib.ptr(a), _DTYPES[a.dtype], T, Z, Y, X, nc, mins.ctypes.data_as(dp),
                                        dens.ctypes.data_as(dp), tab.ctypes.data_as(dp), float(truncate), _MODES[mode],
                                        _lib.ptr(out), out_dtype))
    return out[0] if squeeze_t else out


def normalize(arr, ref=None, channel_normalization="together", eps=1e-8):
    """util/image_processing_3D.py:12-92 (host NumPy; kept for API parity)."""
    arr = np.asarray(arr)
    if arr.ndim in (4, 5):
        mins, dens = _norm_constants(arr, ref, channel_normalization, eps)
        return (arr - mins) / dens if channel_normalization != "separate" else \
            np.stack([(arr[..., c] - mins[c]) / dens[c] for c in range(arr.shape[-1])], -1)
    src = ref if ref is not None else arr
    lo, hi = src.min(), src.max()
    if channel_normalization == "separate":
        rng = hi - lo
        return (arr - lo) / rng if rng > 0 else arr - lo
    return (arr - lo) / (hi - lo + eps)


def apply_gaussian_filter(arr, sigma, mode="reflect", truncate=4.0):
    """util/image_processing_3D.py:95-162 on the device -> float64; ``mode`` is any of scipy.ndimage's boundary modes
    (constant: cval = 0, as scipy.ndimage.gaussian_filter's default)."""
    nc = np.asarray(arr).shape[-1]
    return _run(arr, np.zeros(nc), np.ones(nc), sigma, truncate, False, _lib.F64, mode)


def preprocess_frames(frames, normalization_ref=None, sigma=((1, 1, 1, 0.1), (1, 1, 1, 0.1)),
                      channel_normalization="together", out_float32=False, eps=1e-8):
    """BatchMotionCorrector._preprocess_frames (compensate_recording_3D.py:229-254): normalize then
    Gaussian-filter, fused on the device.  float64 out like the reference (``out_float32`` returns
    what get_displacement will round it to anyway, util/resize_util_3D.py:116)."""
    frames = np.asarray(frames)
    mins, dens = _norm_constants(frames, normalization_ref, channel_normalization, eps)
    return _run(frames, mins, dens, sigma, 4.0, False, _lib.F32 if out_float32 else _lib.F64)
