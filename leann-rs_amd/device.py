"""Minimal HBM buffer wrapper over the C ABI's device helpers (no torch needed)."""
import ctypes as C

import numpy as np

from . import _native as N


class DeviceArray:
    """Typed, shaped device allocation owned by this object."""

    def __init__(self, shape, dtype, device=0):
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.device = device
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        N.check(N.lib().leann_device_malloc(device, max(self.nbytes, 16), C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_host(cls, a, device=0):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype, device)
        if a.nbytes:
            N.check(N.lib().leann_device_upload(d.ptr, a.ctypes.data, a.nbytes))
        return d

    def upload(self, a):
        a = np.ascontiguousarray(a, self.dtype)
        assert a.nbytes == self.nbytes
        if a.nbytes:
            N.check(N.lib().leann_device_upload(self.ptr, a.ctypes.data, a.nbytes))

    def to_host(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            N.check(N.lib().leann_device_download(out.ctypes.data, self.ptr, self.nbytes))
        return out

    def free(self):
        if getattr(self, "ptr", None):
            N.lib().leann_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sync(device=0):
    N.check(N.lib().leann_device_sync(device))
