"""ctypes binding of the C ABI in include/flowreg3d_hip.h (libflowreg3d_hip.so).

The library is plain HIP (no torch types in any signature).  If PyTorch is importable it is
imported *before* the library is loaded so that both share one HIP runtime (torch ships its own
libamdhip64.so with the same soname); torch is never required.

There is no CPU fallback: if the shared library or a GPU is missing every compute entry point
raises ``RuntimeError`` (the executor's ``register()`` then declines, so the reference pipeline
falls back to its own ``sequential3d`` executor -- compensate_recording_3D.py:95-118).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FR3D_LIB") or os.path.join(_HERE, "lib", "libflowreg3d_hip.so")  # FR3D_LIB: A/B builds
MAX_CHANNELS = 8
F32, F64, U8, U16, I16 = 0, 1, 2, 3, 4

K_NAMES = ("sor", "warp", "prefilter", "tensor", "resize", "median", "other", "preproc")


class Params(C.Structure):
    """struct fr3d_params"""
    _fields_ = [
        ("alpha", C.c_double * 3),
        ("update_lag", C.c_int),
        ("iterations", C.c_int),
        ("min_level", C.c_int),
        ("levels", C.c_int),
        ("eta", C.c_double),
        ("a_smooth", C.c_double),
        ("a_data", C.c_double * MAX_CHANNELS),
        ("solver_fp64", C.c_int),
        ("solver_sweep", C.c_int),
        ("reserved", C.c_int * 6),
    ]


class KernelStat(C.Structure):
    """struct fr3d_kernel_stat"""
    _fields_ = [("ms", C.c_double), ("algo_bytes", C.c_double), ("launches", C.c_longlong),
                ("units", C.c_longlong)]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_int, C.c_void_p)

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# every symbol include/flowreg3d_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "fr3d_init": (C.c_int, [C.c_int]),
    "fr3d_shutdown": (None, []),
    "fr3d_last_error": (C.c_char_p, []),
    "fr3d_device_count": (C.c_int, []),
    "fr3d_version": (C.c_char_p, []),
    "fr3d_device_info": (C.c_char_p, []),
    "fr3d_set_batch": (C.c_int, [C.c_int]),
    "fr3d_set_lanes": (C.c_int, [C.c_int]),
    "fr3d_last_solver_mode": (C.c_int, []),
    "fr3d_last_solver_fallback": (C.c_int, []),
    "fr3d_get_displacement": (C.c_int, [C.POINTER(Params), _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "fr3d_get_displacement_dev": (C.c_int, [C.POINTER(Params), _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "fr3d_get_displacement_verify": (C.c_int, [C.POINTER(Params), _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "fr3d_portable_pow": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "fr3d_spline_coefficients": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    "fr3d_motion_tensor_f64": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _vp]),
    "fr3d_level_solve_verify": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_int, _dp,
                                          C.c_double, C.c_double, C.c_double, _vp]),
    "fr3d_level_solve_tensor": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_int, _dp,
                                          C.c_double, C.c_double, C.c_double, C.c_double, _vp]),
    "fr3d_warp": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "fr3d_warp_dev": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "fr3d_process_batch": (C.c_int, [C.POINTER(Params), _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, _vp, _vp, PROGRESS_FN, _vp]),
    "fr3d_process_batch_dev": (C.c_int, [C.POINTER(Params), _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, _vp, _vp, PROGRESS_FN, _vp]),
    "fr3d_process_batch_raw": (C.c_int, [C.POINTER(Params), _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, PROGRESS_FN, _vp]),
    "fr3d_process_batch_raw_dev": (C.c_int, [C.POINTER(Params), _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, PROGRESS_FN, _vp]),
    "fr3d_preprocess": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_double,
                                  _vp, C.c_int]),
    "fr3d_preprocess_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                      C.c_double, _vp, C.c_int]),
    "fr3d_gaussian_filter": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                       C.c_double, C.c_int, _vp, C.c_int]),
    "fr3d_gaussian_filter_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                           C.c_double, C.c_int, _vp, C.c_int]),
    "fr3d_update_reference": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, _dp]),
    "fr3d_update_reference_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, _dp]),
    "fr3d_mean_stack_dev": (C.c_int, [_vp, C.c_int, C.c_size_t, _vp]),
    "fr3d_flow_stats": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _dp]),
    "fr3d_flow_stats_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _dp]),
    "fr3d_resize3d": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "fr3d_resize3d_ex": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _vp]),
    "fr3d_motion_tensor": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _vp, _vp]),
    "fr3d_level_solve": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_int, _dp,
                                   C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, _vp]),
    "fr3d_median5": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    "fr3d_schedule": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _ip, C.c_int, _ip]),
    "fr3d_sor_schedule_check": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.POINTER(C.c_longlong)]),
    "fr3d_dev_malloc": (C.c_void_p, [C.c_size_t]),
    "fr3d_dev_free": (None, [_vp]),
    "fr3d_h2d": (C.c_int, [_vp, _vp, C.c_size_t]),
    "fr3d_d2h": (C.c_int, [_vp, _vp, C.c_size_t]),
    "fr3d_sync": (C.c_int, []),
    "fr3d_prof_enable": (C.c_int, [C.c_int]),
    "fr3d_prof_reset": (C.c_int, []),
    "fr3d_prof_get": (C.c_int, [C.POINTER(KernelStat)]),
    "fr3d_stream_probe": (C.c_int, [C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "fr3d_read_probe": (C.c_int, [C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "fr3d_xcd_probe": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
}

_lib = None
_lock = threading.Lock()
_inited_device = None


def load():
    """dlopen the engine and set the prototypes.  Raises RuntimeError if it is not built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C flowreg3d_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        if "torch" not in sys.modules:
            try:  # share torch's HIP runtime when torch is around (see module docstring)
                import torch  # noqa: F401
            except Exception:
                pass
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("FR3D_LIB") and not hasattr(lib, name):
                continue  # an older build selected for an A/B run may lack newer entry points
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return _lib


def last_error() -> str:
    return load().fr3d_last_error().decode("utf-8", "replace")


def check(rc: int, value_error: bool = False):
    if rc != 0:
        msg = last_error()
        if value_error or msg.startswith("Unsupported interpolation"):
            raise ValueError(msg)
        raise RuntimeError(msg)


def device_count() -> int:
    try:
        return int(load().fr3d_device_count())
    except (RuntimeError, OSError):
        return 0


def init(device: int | None = None):
    """fr3d_init on `device` (default: LOCAL_RANK or 0).  Raises if no GPU is visible."""
    global _inited_device
    lib = load()
    if device is None:
        device = int(os.environ.get("FR3D_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = lib.fr3d_device_count()
        if n > 0:
            device %= n
    if _inited_device == device:
        return lib
    check(lib.fr3d_init(int(device)))
    _inited_device = device
    return lib


def shutdown():
    global _inited_device
    if _lib is not None:
        _lib.fr3d_shutdown()
    _inited_device = None


def ptr(a):
    """void* of a C-contiguous NumPy array (or an int device address, or None)."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous")
    return C.c_void_p(a.ctypes.data)


def make_params(alpha, update_lag, iterations, min_level, levels, eta, a_smooth, a_data, n_channels,
                solver_fp64=None, solver_sweep=0) -> Params:
    p = Params()
    al = np.asarray(alpha, dtype=np.float64).reshape(-1)
    if al.size == 1:
        al = np.repeat(al, 3)
    if al.size != 3:
        raise ValueError("alpha must have 1 or 3 entries")
    for i in range(3):
        p.alpha[i] = float(al[i])
    p.update_lag = int(update_lag)
    p.iterations = int(iterations)
    p.min_level = int(min_level)
    p.levels = int(levels)
    p.eta = float(eta)
    p.a_smooth = float(a_smooth)
    ad = np.broadcast_to(np.asarray(a_data, dtype=np.float64).reshape(-1), (n_channels,)) \
        if np.asarray(a_data).size in (1, n_channels) else None
    if ad is None:
        raise ValueError("a_data must be a scalar or have one entry per channel")
    if n_channels > MAX_CHANNELS:
        raise ValueError(f"at most {MAX_CHANNELS} channels")
    for c in range(n_channels):
        p.a_data[c] = float(ad[c])
    # None = FR3D_SOLVER_AUTO: fp32 solver storage for one channel, fp64 for several
    p.solver_fp64 = -1 if solver_fp64 is None else (int(solver_fp64) if solver_fp64 in (0, 1, 2, 3, True, False) else 1)
    p.solver_sweep = int(solver_sweep)  # 0 = the engine's choice, 1 = plane launches, 2 = window kernel (bit-identical)
    return p


def warn_if_degraded():
    """After a flow solve: FR3D_SOLVER_AUTO fell back from packed to fp32 solver storage for lack of device memory."""
    if load().fr3d_last_solver_fallback():
        import warnings
        warnings.warn("flowreg3d_amd: not enough device memory for packed 42-bit solver storage at this volume size; the "
                      "flow was solved with fp32 storage, which is outside the 1e-4 end-point-error bound against the CPU path "
                      "at 512^3 and beyond (pass solver_fp64=3 to fail instead, or free device memory)", RuntimeWarning,
                      stacklevel=3)


def prof_get() -> dict:
    arr = (KernelStat * len(K_NAMES))()
    check(load().fr3d_prof_get(arr))
    return {K_NAMES[i]: dict(ms=arr[i].ms, algo_bytes=arr[i].algo_bytes, launches=arr[i].launches,
                             units=arr[i].units) for i in range(len(K_NAMES))}
