"""Device memory for tests that hand DEVICE pointers to the C ABI: hipMalloc / hipMemcpy / hipFree of
the HIP runtime the library itself is linked to, through ctypes.  (torch bundles a second copy of
the runtime; whichever of the two is loaded second into a process finds no GPU, so a test that
must run both alone and after the torch-free GPU tests cannot take its memory from torch.)"""
import ctypes as C

import numpy as np

from platypus_amd import native  # noqa: F401  (loads libplatgpu.so and with it the HIP runtime)

_hip = C.CDLL("libamdhip64.so.7")          # by soname: the copy already in the process
_hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
_hip.hipFree.argtypes = [C.c_void_p]
_hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
_hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
_hip.hipDeviceSynchronize.argtypes = []
H2D, D2H = 1, 2


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with HIP error %d" % (what, rc))


class DeviceArray(object):
    """n elements of `dtype` on the current device, filled from `src` or with a byte pattern"""

    def __init__(self, n, dtype, src=None, fill_byte=0xA5):
        self.n, self.dtype = int(n), np.dtype(dtype)
        self.nbytes = max(1, self.n * self.dtype.itemsize)
        p = C.c_void_p()
        _ok(_hip.hipMalloc(C.byref(p), self.nbytes), "hipMalloc")
        self.ptr = p.value
        if src is not None:
            a = np.ascontiguousarray(src, dtype=self.dtype)
            assert len(a) == self.n
            if self.n:
                _ok(_hip.hipMemcpy(self.ptr, a.ctypes.data, a.nbytes, H2D), "hipMemcpy")
        else:
            _ok(_hip.hipMemset(self.ptr, fill_byte, self.nbytes), "hipMemset")
        _ok(_hip.hipDeviceSynchronize(), "hipDeviceSynchronize")

    def numpy(self):
        out = np.empty(self.n, dtype=self.dtype)
        if self.n:
            _ok(_hip.hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, D2H), "hipMemcpy")
        return out

    def free(self):
        if self.ptr:
            _hip.hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]


def free_bytes():
    """free device memory of the current device, as the runtime reports it"""
    free, total = C.c_size_t(0), C.c_size_t(0)
    _ok(_hip.hipMemGetInfo(C.byref(free), C.byref(total)), "hipMemGetInfo")
    return free.value
