"""``HipExecutor3D`` -- the drop-in executor behind flowreg3d's parallelization plugin interface.

Contract mirrored (all under /root/reference/src/flowreg3d/motion_correction/):
  * ``BaseExecutor3D`` (parallelization/base_3d.py:11-120): ``__init__(n_workers)``, ``name``,
    ``process_batch(batch, batch_proc, reference_raw, reference_proc, w_init, get_displacement_func,
    imregister_func, interpolation_method="cubic", progress_callback=None, **kwargs)``
    -> ``(registered (T,Z,Y,X,C) batch.dtype, flow_fields (T,Z,Y,X,3) float32)``, ``setup``,
    ``cleanup``, ``get_info``, context manager, classmethod ``register()``.
  * per-volume body of ``SequentialExecutor3D.process_batch`` (parallelization/sequential_3d.py:
    148-175) -- executed for the whole batch by one C-ABI call, ``fr3d_process_batch``.
  * registry: ``RuntimeContext.register_parallelization_executor(name, cls)`` (_runtime.py:149) with
    name = class name minus "Executor", lower-cased (base_3d.py:97-104) -> ``"hip3d"``; selected by
    ``RegistrationConfig(parallelization="hip")`` (compensate_recording_3D.py:88-94).

When flowreg3d is importable the class registers into its real ``RuntimeContext``; otherwise into the
minimal registry below (same two methods) so the package works standalone on the GPU box.  The two
injected callables are ignored (precedent: MultiprocessingExecutor3D, multiprocessing_3d.py:269-270).
"""
from __future__ import annotations

import ctypes as C
import warnings
from importlib import import_module
from typing import Any, Callable, Dict, Optional, Tuple

import numpy as np

from . import _lib
from .core import _order_of, expand_weight, is_default_weight


_RAW_CODES = {np.dtype(np.float32): _lib.F32, np.dtype(np.float64): _lib.F64, np.dtype(np.uint8): _lib.U8,
              np.dtype(np.uint16): _lib.U16, np.dtype(np.int16): _lib.I16}


class _LocalRuntimeContext:
    """The two registry methods of flowreg3d._runtime.RuntimeContext (:149-199), dotted-path
    storage included, for use when flowreg3d itself is not installed."""
    _config: Dict[str, Any] = {"parallelization_registry": {}, "available_parallelization": set(),
                               "max_workers": 1}

    @classmethod
    def get(cls, key, default=None):
        return cls._config.get(key, default)

    @classmethod
    def register_parallelization_executor(cls, name, executor_class):
        if isinstance(executor_class, type):
            dotted = f"{executor_class.__module__}.{executor_class.__qualname__}"
        else:
            dotted = executor_class
        cls._config["parallelization_registry"][name] = dotted
        cls._config["available_parallelization"].add(name)

    @classmethod
    def get_parallelization_executor(cls, name=None):
        if name is None:
            return None
        dotted = cls._config["parallelization_registry"].get(name)
        if dotted is None:
            return None
        try:
            mod, attr = dotted.rsplit(".", 1)
            return getattr(import_module(mod), attr)
        except (ImportError, AttributeError, ValueError) as e:
            warnings.warn(f"Failed to import executor {name} from {dotted}: {e}")
            return None


def runtime_context():
    """flowreg3d's RuntimeContext when importable, else the local stand-in."""
    try:
        from flowreg3d._runtime import RuntimeContext  # type: ignore
        return RuntimeContext
    except Exception:
        return _LocalRuntimeContext


class HipExecutor3D:
    """Runs the per-volume flow solve + compensation warp of a batch on one MI355X."""

    def __init__(self, n_workers: Optional[int] = None, device: Optional[int] = None):
        self.n_workers = 1  # one process drives one GPU; volumes are pipelined on its stream
        self.name = self.__class__.__name__.replace("Executor", "").lower()
        self.device = device
        self._lib = None

    # -- lifecycle ---------------------------------------------------------------------------
    def setup(self):
        self._lib = _lib.init(self.device)

    def cleanup(self):
        pass  # the engine's workspace is kept for the next batch (same sizes every call)

    def __enter__(self):
        self.setup()
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.cleanup()
        return False

    @classmethod
    def register(cls, force: bool = False) -> bool:
        """Register as ``hip3d``.  Declines (returns False) when no GPU / library is usable so the
        pipeline's fallback chain (compensate_recording_3D.py:95-118) picks ``sequential3d``;
        ``force`` registers regardless (registry tests on machines without a GPU)."""
        if not force and _lib.device_count() < 1:
            return False
        instance_name = cls.__name__.replace("Executor", "").lower()
        runtime_context().register_parallelization_executor(instance_name, cls)
        return True

    def get_info(self) -> Dict[str, Any]:
        return {"name": self.name, "type": self.__class__.__name__, "n_workers": self.n_workers,
                "parallel": False, "backend": "hip/gfx950", "library": _lib.LIB_PATH,
                "description": "MI355X engine: HIP pyramid, B-spline warp, SOR solver, median"}

    # -- the plugin entry point ----------------------------------------------------------------
    def process_batch(self, batch: np.ndarray, batch_proc: np.ndarray, reference_raw: np.ndarray,
                      reference_proc: np.ndarray, w_init: np.ndarray,
                      get_displacement_func: Callable = None, imregister_func: Callable = None,
                      interpolation_method: str = "cubic",
                      progress_callback: Optional[Callable[[int], None]] = None,
                      **kwargs) -> Tuple[np.ndarray, np.ndarray]:
        T, Z, Y, X, nc = batch.shape
        order = _order_of(interpolation_method)
        flow_params_all = dict(kwargs.get("flow_params", {}))
        if flow_params_all.get("cc_initialization", False):
            raise NotImplementedError("cross-correlation pre-alignment is not on the device path "
                                      "(never enabled by the pipeline: compensate_recording_3D.py:301-315)")
        fp = {k: v for k, v in flow_params_all.items() if k not in ("cc_initialization", "cc_hw", "cc_up")}
        # get_displacement's own defaults (core/optical_flow_3d.py:319-333) for missing keys
        alpha = fp.get("alpha", (2, 2, 2))
        a_smooth = float(fp.get("a_smooth", 0.5))
        params = _lib.make_params(alpha, fp.get("update_lag", 10), fp.get("iterations", 20),
                                  fp.get("min_level", 0), fp.get("levels", 50), fp.get("eta", 0.8), a_smooth,
                                  fp.get("a_data", 0.45), nc,
                                  None if fp.get("solver_fp64") is None else int(fp["solver_fp64"]),
                                  int(fp.get("solver_sweep") or 0))
        wt = None if is_default_weight(fp.get("weight", None), nc) else expand_weight(fp.get("weight", None), Z, Y, X, nc)

        def f32(a, shape):
            a = np.ascontiguousarray(a, dtype=np.float32)
            if a.shape != shape:
                raise ValueError(f"array has shape {a.shape}, expected {shape}")
            return a

        bp = f32(batch_proc, (T, Z, Y, X, nc))  # the reference casts to fp32 itself (util/resize_util_3D.py:116)
        rp = f32(np.asarray(reference_proc).reshape(Z, Y, X, nc), (Z, Y, X, nc))
        wi = None if w_init is None else f32(w_init, (Z, Y, X, 3))
        w32 = None if wt is None else f32(wt, (Z, Y, X, nc))
        # The final warp works on the RAW volume in its own dtype (sequential_3d.py:153-170): SciPy builds
        # the spline from the raw values and allocates map_coordinates' output in the input's dtype, so
        # integer batches are rounded and saturated, not truncated.  float32 / float64 / uint8 / uint16 /
        # int16 go to the device as they are; anything else is widened to float64 first.
        raw_code = _RAW_CODES.get(batch.dtype)
        br = np.ascontiguousarray(batch if raw_code is not None else batch.astype(np.float64))
        if br.shape != (T, Z, Y, X, nc):
            raise ValueError(f"array has shape {br.shape}, expected {(T, Z, Y, X, nc)}")
        code = raw_code if raw_code is not None else _lib.F64
        rr = np.asarray(reference_raw).reshape(Z, Y, X, nc)
        rr = np.ascontiguousarray(rr, dtype=np.float32 if rr.dtype == np.float32 else np.float64)
        ref_code = _lib.F32 if rr.dtype == np.float32 else _lib.F64
        flows = np.empty((T, Z, Y, X, 3), np.float32)
        registered = np.empty_like(batch)  # np.empty_like(batch), sequential_3d.py:75
        direct = raw_code is not None and registered.flags.c_contiguous
        reg_dev = registered if direct else np.empty((T, Z, Y, X, nc), br.dtype)

        cb_error = []

        def _progress(n, _user):
            if progress_callback is not None:
                try:
                    progress_callback(int(n))
                except Exception as e:  # never unwind through the C frame
                    cb_error.append(e)

        cb = _lib.PROGRESS_FN(_progress)
        lib = self._lib or _lib.init(self.device)
        _lib.check(lib.fr3d_process_batch_raw(C.byref(params), _lib.ptr(bp), _lib.ptr(br), code, _lib.ptr(rp),
                                              _lib.ptr(rr), ref_code, _lib.ptr(wi), _lib.ptr(w32), T, Z, Y, X, nc, order,
                                              _lib.ptr(flows), _lib.ptr(reg_dev), cb, None))
        _lib.warn_if_degraded()
        if cb_error:
            raise cb_error[0]
        if not direct:
            registered[...] = reg_dev
        return registered, flows

    # -- sharded series: reference payload already in HBM (flowreg3d_amd/distributed.py) -------------------
    def process_batch_device_refs(self, batch: np.ndarray, batch_proc: np.ndarray, ref_ptrs: Dict[str, Optional[int]],
                                  ref_shape: Tuple[int, int, int, int], interpolation_method: str = "cubic",
                                  progress_callback: Optional[Callable[[int], None]] = None, flow_params=None
                                  ) -> Tuple[np.ndarray, np.ndarray]:
        """``process_batch`` with reference_raw / reference_proc / w_init / weight given as DEVICE addresses of
        float32 arrays (the fields of the broadcast buffer): only this rank's volumes cross PCIe."""
        from .device import DeviceBuffer
        T, Z, Y, X, nc = batch.shape
        if (Z, Y, X, nc) != tuple(ref_shape):
            raise ValueError(f"batch volumes {(Z, Y, X, nc)} do not match the reference {tuple(ref_shape)}")
        order = _order_of(interpolation_method)
        fp = {k: v for k, v in dict(flow_params or {}).items() if k not in ("cc_initialization", "cc_hw", "cc_up", "weight")}
        params = _lib.make_params(fp.get("alpha", (2, 2, 2)), fp.get("update_lag", 10), fp.get("iterations", 20),
                                  fp.get("min_level", 0), fp.get("levels", 50), fp.get("eta", 0.8),
                                  float(fp.get("a_smooth", 0.5)), fp.get("a_data", 0.45), nc,
                                  None if fp.get("solver_fp64") is None else int(fp["solver_fp64"]),
                                  int(fp.get("solver_sweep") or 0))
        raw_code = _RAW_CODES.get(batch.dtype)
        br = np.ascontiguousarray(batch if raw_code is not None else batch.astype(np.float64))
        code = raw_code if raw_code is not None else _lib.F64
        lib = self._lib or _lib.init(self.device)
        bufs = []
        try:
            d_proc = DeviceBuffer((T, Z, Y, X, nc), np.float32).upload(batch_proc)
            bufs.append(d_proc)
            d_raw = DeviceBuffer((T, Z, Y, X, nc), br.dtype).upload(br)
            bufs.append(d_raw)
            d_flows = DeviceBuffer((T, Z, Y, X, 3), np.float32)
            bufs.append(d_flows)
            d_reg = DeviceBuffer((T, Z, Y, X, nc), br.dtype)
            bufs.append(d_reg)
            cb_error = []

            def _progress(n, _user):
                if progress_callback is not None:
                    try:
                        progress_callback(int(n))
                    except Exception as e:
                        cb_error.append(e)

            cb = _lib.PROGRESS_FN(_progress)
            _lib.check(lib.fr3d_process_batch_raw_dev(C.byref(params), d_proc.ptr, d_raw.ptr, code, ref_ptrs["reference_proc"],
                                                      ref_ptrs["reference_raw"], _lib.F32, ref_ptrs.get("w_init"),
                                                      ref_ptrs.get("weight"), T, Z, Y, X, nc, order, d_flows.ptr, d_reg.ptr,
                                                      cb, None))
            _lib.warn_if_degraded()
            if cb_error:
                raise cb_error[0]
            flows = d_flows.download()
            reg = d_reg.download()
        finally:
            for b in bufs:
                b.free()
        registered = np.empty_like(batch)
        registered[...] = reg
        return registered, flows

