"""Device-resident moment arrays for hosts without a HIP array type (plain ctypes + numpy).

Layout: shape (planes, n) row-major == moment-major SoA planes with ld = n, the layout of Julia's
column-major m[parcel, moment] (reference test/examples/utils/rainshaft_helpers.jl:48-56).
"""
import ctypes as C

import numpy as np

from . import _lib


class DeviceArray:
    """A float64 (or float32, for CLOUDY_F32 plans) (planes, n) buffer in HBM owned through cloudy_malloc/cloudy_free."""

    def __init__(self, planes, n, dtype=np.float64):
        self.shape = (int(planes), int(n))
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise TypeError("DeviceArray holds float64 or float32")
        self.nbytes = self.dtype.itemsize * self.shape[0] * self.shape[1]
        p = C.c_void_p()
        _lib.check(_lib.lib().cloudy_malloc(C.byref(p), max(self.nbytes, 8)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a):
        a = np.asarray(a)
        a = np.ascontiguousarray(a, dtype=np.float32 if a.dtype == np.float32 else np.float64)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        d = cls(*a.shape, dtype=a.dtype)
        if d.nbytes:
            _lib.check(_lib.lib().cloudy_memcpy_h2d(d.ptr, a.ctypes.data, d.nbytes, None))
            _lib.check(_lib.lib().cloudy_stream_synchronize(None))
        return d

    @classmethod
    def zeros(cls, planes, n, dtype=np.float64):
        d = cls(planes, n, dtype)
        _lib.check(_lib.lib().cloudy_memset(d.ptr, 0, d.nbytes, None))
        return d

    def to_numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            _lib.check(_lib.lib().cloudy_stream_synchronize(None))
            _lib.check(_lib.lib().cloudy_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes, None))
            _lib.check(_lib.lib().cloudy_stream_synchronize(None))
        return out

    def columns_to_numpy(self, ncols, lo=0):
        """parcels [lo, lo + ncols) of every plane (one copy per plane): diagnostics on a slice of a big result"""
        lo = int(lo)
        ncols = max(0, min(int(ncols), self.shape[1] - lo))
        out = np.empty((self.shape[0], ncols), dtype=self.dtype)
        L = _lib.lib()
        _lib.check(L.cloudy_stream_synchronize(None))
        for q in range(self.shape[0]):
            _lib.check(L.cloudy_memcpy_d2h(out[q].ctypes.data, self.ptr + (q * self.shape[1] + lo) * self.dtype.itemsize,
                                           ncols * self.dtype.itemsize, None))
        _lib.check(L.cloudy_stream_synchronize(None))
        return out

    def set_columns(self, lo, a):
        """upload the host block a (planes, m) into parcels [lo, lo + m) of every plane (the shard of one rank inside a
        batch that holds every rank's parcels)"""
        a = np.ascontiguousarray(a, dtype=self.dtype)
        lo = int(lo)
        if a.shape[0] != self.shape[0] or lo < 0 or lo + a.shape[1] > self.shape[1]:
            raise ValueError("block does not fit")
        L = _lib.lib()
        for q in range(self.shape[0]):
            _lib.check(L.cloudy_memcpy_h2d(self.ptr + (q * self.shape[1] + lo) * self.dtype.itemsize, a[q].ctypes.data,
                                           a.shape[1] * self.dtype.itemsize, None))
        _lib.check(L.cloudy_stream_synchronize(None))

    def data_ptr(self):
        return self.ptr

    @property
    def ld(self):
        return self.shape[1]

    def free(self):
        if getattr(self, "ptr", None):
            _lib.lib().cloudy_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def plane_dtype(plan):
    """numpy element type of the moment planes a plan reads and writes: float32 for CLOUDY_F32 / CLOUDY_F32_FAST, float64
    for CLOUDY_F64 and CLOUDY_F64_RELAXED (dtype code 3: fp64 planes, relaxed series tolerance -- include/cloudy_hip.h).
    The one place the Python mirror derives it (`plan.dtype >= 1` sized float32 buffers for the relaxed dtype)."""
    return np.dtype(np.float32) if int(plan.dtype) in (1, 2) else np.dtype(np.float64)


def dtype_code(x):
    """CLOUDY_F64 (0) / CLOUDY_F32 (1) of a device array."""
    if isinstance(x, DeviceArray):
        return 1 if x.dtype == np.float32 else 0
    return 1 if x.dtype.itemsize == 4 else 0


def as_device(x):
    """(pointer, planes, n, ld) of a DeviceArray or of a CUDA/HIP torch tensor of shape (planes, n)."""
    if isinstance(x, DeviceArray):
        return x.ptr, x.shape[0], x.shape[1], x.shape[1]
    if hasattr(x, "data_ptr") and hasattr(x, "is_cuda"):
        if not x.is_cuda:
            raise TypeError("torch tensor must live on the GPU (no CPU fallback)")
        if x.dtype.itemsize not in (4, 8) or not x.dtype.is_floating_point:
            raise TypeError("expected a float64 (or float32) tensor")
        if x.dim() != 2 or x.stride(1) != 1:
            raise TypeError("expected a (planes, n) tensor with contiguous parcels")
        return x.data_ptr(), x.shape[0], x.shape[1], x.stride(0) if x.shape[0] > 1 else x.shape[1]
    raise TypeError(f"unsupported device array type {type(x)!r}")
