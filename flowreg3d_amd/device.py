"""Device-resident buffers and the output sink of a registered series.

The reference streams ``registered`` and ``w`` of every batch to file writers
(motion_correction/compensate_recording_3D.py:172-196, 510-523).  File formats are out of scope; what
the device path needs from that stage is a place where the outputs of a long series can stay in HBM
between batches (288 GB per GPU) and be fetched -- whole, by time range, or not at all -- when a
consumer wants them.  ``DeviceSink`` is that place; ``BatchMotionCorrectorHip.run(..., sink="device")``
fills it without a host round trip of the outputs.
"""
from __future__ import annotations

import numpy as np

from . import _lib


class DeviceBuffer:
    """A typed array in HBM, owned through the engine's allocator (fr3d_dev_malloc / fr3d_dev_free)."""

    def __init__(self, shape, dtype):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._lib = _lib.init()
        self.ptr = self._lib.fr3d_dev_malloc(max(self.nbytes, 1))
        if not self.ptr:
            raise MemoryError(f"fr3d_dev_malloc({self.nbytes}) failed: {_lib.last_error()}")

    def at(self, index0: int) -> int:
        """device address of element [index0, 0, ...]"""
        stride = self.nbytes // self.shape[0] if self.shape and self.shape[0] else 0
        return self.ptr + int(index0) * stride

    def upload(self, a, index0: int = 0):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.nbytes + (self.at(index0) - self.ptr) > self.nbytes:
            raise ValueError("upload past the end of the device buffer")
        _lib.check(self._lib.fr3d_h2d(self.at(index0), a.ctypes.data, a.nbytes))
        return self

    def download(self, t0: int = 0, t1=None) -> np.ndarray:
        t1 = self.shape[0] if t1 is None else int(t1)
        out = np.empty((max(t1 - t0, 0),) + self.shape[1:], self.dtype)
        if out.nbytes:
            _lib.check(self._lib.fr3d_d2h(out.ctypes.data, self.at(t0), out.nbytes))
        return out

    def download_into(self, out: np.ndarray, t0: int = 0):
        """elements [t0, t0 + len(out)) straight into a C-contiguous host array (or slice) of this buffer's dtype"""
        if out.dtype != self.dtype or not out.flags["C_CONTIGUOUS"] or out.shape[1:] != self.shape[1:]:
            raise ValueError("download_into needs a C-contiguous array of the buffer's dtype and trailing shape")
        if t0 + out.shape[0] > self.shape[0]:
            raise ValueError("download past the end of the device buffer")
        if out.nbytes:
            _lib.check(self._lib.fr3d_d2h(out.ctypes.data, self.at(t0), out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self._lib.fr3d_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):  # best effort; free() is the explicit way
        try:
            self.free()
        except Exception:
            pass


class DeviceSink:
    """``registered`` (T,Z,Y,X,C) in the raw dtype and ``w`` (T,Z,Y,X,3) float32 of a series, kept in HBM."""

    def __init__(self, T, Z, Y, X, C, raw_dtype):
        self.registered_dev = DeviceBuffer((T, Z, Y, X, C), raw_dtype)
        self.flows_dev = DeviceBuffer((T, Z, Y, X, 3), np.float32)
        self.filled = 0

    @property
    def nbytes(self) -> int:
        return self.registered_dev.nbytes + self.flows_dev.nbytes

    def registered(self, t0: int = 0, t1=None) -> np.ndarray:
        return self.registered_dev.download(t0, self.filled if t1 is None else t1)

    def flows(self, t0: int = 0, t1=None) -> np.ndarray:
        return self.flows_dev.download(t0, self.filled if t1 is None else t1)

    def free(self):
        self.registered_dev.free()
        self.flows_dev.free()
