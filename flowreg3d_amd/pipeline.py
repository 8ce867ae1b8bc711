"""Batch driver around the executor: what
``BatchMotionCorrector.run`` (motion_correction/compensate_recording_3D.py:431-555) and
``compensate_arr_3D`` (motion_correction/compensate_arr_3D.py:13-143) do for in-memory arrays,
with the per-voxel work on the device:

  reference: weights (:212-224), reference_proc = preprocess(reference_raw) (:227)
  per batch of ``buffer_size`` volumes: preprocess against the reference's range (:461-463),
  first batch: w_init = mean flow of the first min(22,T) volumes solved from zero (:342-393),
  executor.process_batch (:476-478), w_init <- mean of the last <= 20 flows (:481-485),
  statistics mean/max |w|, mean divergence, mean translation (:488-508).

The file formats, readers/writers and the pydantic ``OFOptions`` model are out of scope; ``Options``
below carries the fields of OFOptions this driver reads (OF_options_3D.py:155-231) with the same
defaults, and any object with those attributes (an actual OFOptions included) is accepted.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Any, Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .executor import HipExecutor3D
from .preprocess import preprocess_frames

_OUT_DTYPES = {"single": np.float32, "double": np.float64, "uint8": np.uint8, "uint16": np.uint16,
               "int16": np.int16, "int32": np.int32}


@dataclass
class Options:
    """Subset of OFOptions (motion_correction/OF_options_3D.py:155-231), same defaults."""
    alpha: Any = (0.25, 0.25, 0.25)
    weight: Any = field(default_factory=lambda: [0.5, 0.5])
    levels: int = 100
    min_level: int = 5
    quality_setting: str = "quality"
    eta: float = 0.8
    update_lag: int = 5
    iterations: int = 100
    a_smooth: float = 1.0
    a_data: float = 0.45
    sigma: Any = field(default_factory=lambda: [[1.0, 1.0, 1.0, 0.1], [1.0, 1.0, 1.0, 0.1]])
    buffer_size: int = 10
    output_typename: Optional[str] = "double"
    channel_normalization: str = "together"
    interpolation_method: str = "cubic"
    update_initialization_w: bool = True
    update_reference: bool = False
    solver_fp64: Optional[int] = None  # extension: None = the library's choice (fr3d_params.solver_fp64: by size and channel count)

    @property
    def effective_min_level(self) -> int:
        """OF_options_3D.py:329-341"""
        if self.min_level >= 0:
            return self.min_level
        return {"quality": 0, "balanced": 4, "fast": 6}.get(str(self.quality_setting), 0)


def _alpha3(alpha):
    """OF_options_3D.py:239-264: scalar / 2-tuple / 3-tuple -> 3-tuple."""
    a = np.asarray(alpha, dtype=np.float64).reshape(-1)
    if a.size == 1:
        return (float(a[0]),) * 3
    if a.size == 2:
        return (float(a[0]), float(a[1]), float(a[1]))
    if a.size == 3:
        return tuple(float(x) for x in a)
    raise ValueError("alpha must have 1, 2 or 3 entries")


def _weight_at(weight, i, n_channels):
    """OFOptions.get_weight_at (OF_options_3D.py:371-399)."""
    w = np.asarray(weight, dtype=float)
    if w.ndim <= 1:
        if w.size == 1:
            return float(w.reshape(-1)[0])
        if w.size > n_channels:
            w = w[:n_channels]
            w = w / w.sum()
        if i >= w.size:
            return 1.0 / n_channels
        return float(w[i])
    if i >= w.shape[0]:
        return np.ones(w.shape[1:]) / n_channels
    return w[i]


def _opt(options, name, default):
    v = getattr(options, name, default)
    return getattr(v, "value", v)  # enums of the real OFOptions


@dataclass
class BatchStats:
    mean_disp: List[float] = field(default_factory=list)
    max_disp: List[float] = field(default_factory=list)
    mean_div: List[float] = field(default_factory=list)
    mean_translation: List[float] = field(default_factory=list)


def flow_statistics(w: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Per-volume (mean |w|, max |w|, mean divergence, |mean translation|) on the device
    (compensate_recording_3D.py:488-508)."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    T, Z, Y, X, _ = w.shape
    out = np.zeros((T, 6), np.float64)
    lib = _lib.init()
    _lib.check(lib.fr3d_flow_stats(_lib.ptr(w), T, Z, Y, X, out.ctypes.data_as(C.POINTER(C.c_double))))
    trans = np.sqrt(out[:, 3] ** 2 + out[:, 4] ** 2 + out[:, 5] ** 2)
    return out[:, 0], out[:, 1], out[:, 2], trans


def update_reference(batch_proc: np.ndarray, w: np.ndarray, reference_proc: np.ndarray,
                     interpolation_method: str = "cubic") -> np.ndarray:
    """``BatchMotionCorrector._update_reference`` (compensate_recording_3D.py:395-429) on the device: per
    channel the last <= 100 ``batch_proc`` volumes are warped by their flows ``w`` (T,Z,Y,X,3) and averaged in
    float64 -> new ``reference_proc`` (Z,Y,X,C) float64.  An empty batch returns the reference unchanged."""
    from .core import _order_of
    bp = np.asarray(batch_proc)
    bp = np.ascontiguousarray(bp, dtype=np.float32 if bp.dtype == np.float32 else np.float64)
    rp = np.asarray(reference_proc)
    rp = np.ascontiguousarray(rp, dtype=np.float32 if rp.dtype == np.float32 else np.float64)
    fl = np.ascontiguousarray(w, dtype=np.float32)
    T, Z, Y, X, nc = bp.shape
    if rp.shape != (Z, Y, X, nc) or fl.shape != (T, Z, Y, X, 3):
        raise ValueError(f"incompatible shapes {bp.shape} / {fl.shape} / {rp.shape}")
    out = rp.astype(np.float64)  # a copy; stays as it is when T == 0
    lib = _lib.init()
    _lib.check(lib.fr3d_update_reference(_lib.ptr(bp), _lib.F32 if bp.dtype == np.float32 else _lib.F64, _lib.ptr(fl),
                                         _lib.ptr(rp), _lib.F32 if rp.dtype == np.float32 else _lib.F64, T, Z, Y, X, nc,
                                         _order_of(interpolation_method), out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


class BatchMotionCorrectorHip:
    """In-memory counterpart of BatchMotionCorrector for (T,Z,Y,X,C) arrays."""

    def __init__(self, options: Any = None, executor=None):
        self.options = options if options is not None else Options()
        self.executor = executor if executor is not None else HipExecutor3D()
        self.stats = BatchStats()
        self.w_init: Optional[np.ndarray] = None
        self._callbacks: List[Callable[[int, int], None]] = []

    def register_progress_callback(self, cb: Callable[[int, int], None]):
        self._callbacks.append(cb)

    # -- pieces of BatchMotionCorrector ------------------------------------------------------
    def _flow_params(self) -> dict:
        o = self.options
        return {"alpha": _alpha3(_opt(o, "alpha", (0.25,) * 3)), "weight": self.weight,
                "levels": int(_opt(o, "levels", 100)),
                "min_level": int(getattr(o, "effective_min_level", getattr(o, "min_level", 0))),
                "eta": float(_opt(o, "eta", 0.8)), "update_lag": int(_opt(o, "update_lag", 5)),
                "iterations": int(_opt(o, "iterations", 100)), "a_smooth": float(_opt(o, "a_smooth", 1.0)),
                "a_data": _opt(o, "a_data", 0.45), "solver_fp64": _opt(o, "solver_fp64", None)}

    def _preprocess(self, frames, normalization_ref=None):
        return preprocess_frames(frames, normalization_ref=normalization_ref, sigma=np.asarray(_opt(self.options, "sigma", None)),
                                 channel_normalization=str(_opt(self.options, "channel_normalization", "together")))

    def _setup_reference(self, reference: np.ndarray):
        self.reference_raw = np.asarray(reference).astype(np.float64)
        Z, Y, X = self.reference_raw.shape[:3]
        nc = self.reference_raw.shape[3]
        self.weight = np.ones((Z, Y, X, nc), np.float64)
        for c in range(nc):
            self.weight[..., c] = _weight_at(_opt(self.options, "weight", [1.0] * nc), c, nc)
        self.reference_proc = self._preprocess(self.reference_raw)

    def _process(self, batch, batch_proc, w_init, notify):
        cb = None
        if notify and self._callbacks:
            def cb(n):
                self._done += int(n)
                for f in self._callbacks:
                    f(self._done, self._total)
        return self.executor.process_batch(batch, batch_proc, self.reference_raw, self.reference_proc, w_init,
                                           None, None,
                                           interpolation_method=str(_opt(self.options, "interpolation_method", "cubic")),
                                           progress_callback=cb, flow_params=self._flow_params())

    def run(self, video: np.ndarray, reference: np.ndarray, sink: str = "host"):
        """-> (registered, flows) as NumPy arrays (``sink="host"``), or a ``DeviceSink`` holding both in HBM
        (``sink="device"``: batches are uploaded once, preprocessing, flow, warp, w_init updates, statistics and
        the optional reference update all run on device-resident data)."""
        if sink == "device":
            return self._run_device(np.asarray(video), reference)
        if sink not in ("host", "host_arrays"):
            raise ValueError("sink must be 'host', 'host_arrays' or 'device'")
        video = np.asarray(video)
        from .executor import _RAW_CODES
        if sink == "host" and video.dtype in _RAW_CODES and video.ndim == 5 and type(self.executor) is HipExecutor3D:
            # the same driver on device-resident batches: the raw batch goes up once, preprocessing / flow / warp /
            # w_init means / statistics run in HBM, and only `registered` and `w` of each batch come back (the float64
            # `batch_proc` of the array path below never crosses PCIe: 4x the raw volume each way)
            # ("host_arrays": the array-at-a-time path below, what a custom executor or another dtype takes)
            return self._run_device(video, reference, host=True)
        T = video.shape[0]
        self._total, self._done = T, 0
        self._setup_reference(reference)
        Z, Y, X, nc = self.reference_raw.shape
        registered = np.empty_like(video)
        flows = np.empty((T, Z, Y, X, 3), np.float32)
        bs = max(1, int(_opt(self.options, "buffer_size", 10)))
        self.executor.setup()
        try:
            for bi, t0 in enumerate(range(0, T, bs)):
                batch = video[t0:t0 + bs]
                batch_proc = self._preprocess(batch, normalization_ref=self.reference_raw)
                if bi == 0:
                    # w_init from the first min(22, T) volumes solved from zero (:342-393)
                    n_init = min(22, batch.shape[0])
                    _, w0 = self._process(batch[:n_init], batch_proc[:n_init], np.zeros((Z, Y, X, 3), np.float32), False)
                    self.w_init = np.mean(w0, axis=0)
                use_init = bool(_opt(self.options, "update_initialization_w", True))
                cur = self.w_init if use_init else np.zeros_like(self.w_init)
                reg, w = self._process(batch, batch_proc, cur, True)
                if use_init:
                    self.w_init = np.mean(w[-20:], axis=0) if w.shape[0] > 20 else np.mean(w, axis=0)
                md, mx, dv, tr = flow_statistics(w)
                self.stats.mean_disp.extend(md.tolist())
                self.stats.max_disp.extend(mx.tolist())
                self.stats.mean_div.extend(dv.tolist())
                self.stats.mean_translation.extend(tr.tolist())
                registered[t0:t0 + bs] = reg
                flows[t0:t0 + bs] = w
                if bool(_opt(self.options, "update_reference", False)):  # compensate_recording_3D.py:525-526
                    self.reference_proc = update_reference(batch_proc, w, self.reference_proc,
                                                           str(_opt(self.options, "interpolation_method", "cubic")))
        finally:
            self.executor.cleanup()
        return registered, flows

    def _run_device(self, video: np.ndarray, reference: np.ndarray, host: bool = False):
        """The same driver with every per-voxel array resident in HBM (same arithmetic, same call order).
        ``host``: the outputs of each batch are fetched into NumPy arrays (-> registered, flows) and the device holds one
        batch of them; otherwise the whole series' outputs stay in a DeviceSink."""
        from .device import DeviceBuffer, DeviceSink
        from .executor import _RAW_CODES
        from .preprocess import _DTYPES, _norm_constants, _sigma_table
        from .core import _order_of, expand_weight
        if video.dtype not in _RAW_CODES:
            raise TypeError(f"device sink supports {sorted(str(d) for d in _RAW_CODES)} series, got {video.dtype}")
        T = video.shape[0]
        self._total, self._done = T, 0
        self._setup_reference(reference)
        Z, Y, X, nc = self.reference_raw.shape
        nv = Z * Y * X
        lib = _lib.init(getattr(self.executor, "device", None))
        dp = C.POINTER(C.c_double)
        fp = self._flow_params()
        params = _lib.make_params(fp["alpha"], fp["update_lag"], fp["iterations"], fp["min_level"], fp["levels"], fp["eta"],
                                  fp["a_smooth"], fp["a_data"], nc, fp["solver_fp64"])
        order = _order_of(str(_opt(self.options, "interpolation_method", "cubic")))
        bs = max(1, int(_opt(self.options, "buffer_size", 10)))
        upd_ref = bool(_opt(self.options, "update_reference", False))
        use_init = bool(_opt(self.options, "update_initialization_w", True))
        mins, dens = _norm_constants(video[:1], self.reference_raw, str(_opt(self.options, "channel_normalization", "together")), 1e-8)
        mins = np.ascontiguousarray(mins, np.float64)
        dens = np.ascontiguousarray(dens, np.float64)
        tab = _sigma_table(np.asarray(_opt(self.options, "sigma", None)), nc)
        sink = DeviceSink(min(bs, T) if host else T, Z, Y, X, nc, video.dtype)
        registered = np.empty_like(video) if host else None
        flows = np.empty((T, Z, Y, X, 3), np.float32) if host else None
        bufs = []

        def dev(shape, dtype, init=None):
            b = DeviceBuffer(shape, dtype)
            bufs.append(b)
            return b.upload(init) if init is not None else b

        try:
            raw = dev((bs, Z, Y, X, nc), video.dtype)
            proc = dev((bs, Z, Y, X, nc), np.float32)
            proc64 = dev((bs, Z, Y, X, nc), np.float64) if upd_ref else None
            ref_raw = dev((Z, Y, X, nc), np.float64, self.reference_raw)
            ref_proc = dev((Z, Y, X, nc), np.float32, self.reference_proc)
            ref_proc64 = dev((Z, Y, X, nc), np.float64, self.reference_proc) if upd_ref else None
            new_ref64 = dev((Z, Y, X, nc), np.float64) if upd_ref else None
            weight = dev((Z, Y, X, nc), np.float32, expand_weight(self.weight, Z, Y, X, nc))
            w_init = dev((Z, Y, X, 3), np.float32, np.zeros((Z, Y, X, 3), np.float32))
            zero = dev((Z, Y, X, 3), np.float32, np.zeros((Z, Y, X, 3), np.float32))
            stats = np.zeros((bs, 6), np.float64)

            def process(n, init_ptr, flows_ptr, reg_ptr, notify):
                done = []
                cb = _lib.PROGRESS_FN((lambda k, _u: done.append(int(k))) if notify else (lambda k, _u: None))
                _lib.check(lib.fr3d_process_batch_raw_dev(C.byref(params), proc.ptr, raw.ptr, _RAW_CODES[video.dtype], ref_proc.ptr,
                                                          ref_raw.ptr, _lib.F64, init_ptr, weight.ptr, n, Z, Y, X, nc, order,
                                                          flows_ptr, reg_ptr, cb, None))
                if notify and self._callbacks:
                    for k in done:
                        self._done += k
                        for f in self._callbacks:
                            f(self._done, self._total)

            for bi, t0 in enumerate(range(0, T, bs)):
                n = min(bs, T - t0)
                raw.upload(video[t0:t0 + n])
                _lib.check(lib.fr3d_preprocess_dev(raw.ptr, _DTYPES[video.dtype], n, Z, Y, X, nc, mins.ctypes.data_as(dp),
                                                   dens.ctypes.data_as(dp), tab.ctypes.data_as(dp), 4.0, proc.ptr, _lib.F32))
                if upd_ref:
                    _lib.check(lib.fr3d_preprocess_dev(raw.ptr, _DTYPES[video.dtype], n, Z, Y, X, nc, mins.ctypes.data_as(dp),
                                                       dens.ctypes.data_as(dp), tab.ctypes.data_as(dp), 4.0, proc64.ptr, _lib.F64))
                o0 = 0 if host else t0  # where this batch's outputs sit in the sink
                fl_ptr, reg_ptr = sink.flows_dev.at(o0), sink.registered_dev.at(o0)
                if bi == 0:  # w_init = mean flow of the first min(22, T) volumes solved from zero (:342-393)
                    n_init = min(22, n)
                    process(n_init, zero.ptr, fl_ptr, reg_ptr, False)
                    _lib.check(lib.fr3d_mean_stack_dev(fl_ptr, n_init, nv * 3, w_init.ptr))
                process(n, w_init.ptr if use_init else zero.ptr, fl_ptr, reg_ptr, True)
                if use_init:  # mean of the last <= 20 flows of the batch (:481-485)
                    k = min(n, 20)
                    _lib.check(lib.fr3d_mean_stack_dev(sink.flows_dev.at(o0 + n - k), k, nv * 3, w_init.ptr))
                _lib.check(lib.fr3d_flow_stats_dev(fl_ptr, n, Z, Y, X, stats.ctypes.data_as(dp)))
                self.stats.mean_disp.extend(stats[:n, 0].tolist())
                self.stats.max_disp.extend(stats[:n, 1].tolist())
                self.stats.mean_div.extend(stats[:n, 2].tolist())
                self.stats.mean_translation.extend(np.sqrt(stats[:n, 3] ** 2 + stats[:n, 4] ** 2 + stats[:n, 5] ** 2).tolist())
                if upd_ref:  # compensate_recording_3D.py:525-526, 395-429
                    _lib.check(lib.fr3d_update_reference_dev(proc64.ptr, _lib.F64, fl_ptr, ref_proc64.ptr, _lib.F64, n, Z, Y, X,
                                                             nc, order, C.cast(C.c_void_p(new_ref64.ptr), dp)))
                    self.reference_proc = new_ref64.download()
                    ref_proc64.upload(self.reference_proc)
                    ref_proc.upload(self.reference_proc)
                if host:
                    sink.registered_dev.download_into(registered[t0:t0 + n])
                    sink.flows_dev.download_into(flows[t0:t0 + n])
                sink.filled = n if host else t0 + n
            self.w_init = w_init.download()
        except Exception:
            sink.free()
            raise
        finally:
            for b in bufs:
                b.free()
        if host:
            sink.free()
            return registered, flows
        return sink


def compensate_arr_3D(c1: np.ndarray, c_ref: np.ndarray, options: Any = None,
                      progress_callback: Optional[Callable[[int, int], None]] = None,
                      return_stats: bool = False):
    """motion_correction/compensate_arr_3D.py:13-143 on the MI355X engine.

    c1: (T,Z,Y,X,C), (T,Z,Y,X) with a 3-D reference, or a single (Z,Y,X) volume; c_ref: (Z,Y,X[,C]).
    Returns (c_reg with the input's shape, w (T,Z,Y,X,3) float32 [squeezed like the reference])."""
    c1 = np.asarray(c1)
    c_ref = np.asarray(c_ref)
    if c1.size == 0:
        raise ValueError("Input array cannot be empty")
    squeezed = False
    original_shape = c1.shape
    if c1.ndim == 4 and c_ref.ndim == 3:
        c1 = c1[..., np.newaxis]
        c_ref = c_ref[..., np.newaxis]
        squeezed = True
    elif c1.ndim == 3:
        c1 = c1[np.newaxis, :, :, :, np.newaxis]
        if c_ref.ndim == 3:
            c_ref = c_ref[..., np.newaxis]
        squeezed = True
    if c1.ndim != 5 or c_ref.ndim != 4 or c1.shape[1:] != c_ref.shape:
        raise ValueError(f"incompatible shapes {original_shape} / {c_ref.shape}")
    corrector = BatchMotionCorrectorHip(options)
    if progress_callback is not None:
        corrector.register_progress_callback(progress_callback)
    c_reg, w = corrector.run(c1, c_ref)
    typename = _opt(corrector.options, "output_typename", None)
    if typename in _OUT_DTYPES:
        c_reg = c_reg.astype(_OUT_DTYPES[typename])
    if squeezed:
        if len(original_shape) == 3:
            c_reg = np.squeeze(c_reg)
            w = np.squeeze(w, axis=0)
        else:
            c_reg = c_reg[..., 0]
    return (c_reg, w, corrector.stats) if return_stats else (c_reg, w)
