"""Device buffers for tests through the HIP runtime directly (ctypes) -- TEST INFRASTRUCTURE.  The library under test
takes plain device pointers; going through torch for a few message buffers would initialise a second GPU client in
the test process for nothing."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        # the HIP runtime the library under test is linked against: the copy already mapped into this process
        from fdtd2d_amd import _abi
        _abi.load()
        # (by SONAME: dlopen returns the handle of the copy that is already loaded under that name)
        for name in ["libamdhip64.so.7", "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"]:
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise ImportError("libamdhip64.so not found")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    return _hip


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with hipError {rc}")


def sync():
    _ck(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")


class DevBuf:
    """nbytes of device memory, zeroed."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _ck(hip().hipMalloc(C.byref(p), self.nbytes), "hipMalloc")
        self.ptr = p.value
        _ck(hip().hipMemset(self.ptr, 0, self.nbytes), "hipMemset")

    def upload(self, a):
        a = np.ascontiguousarray(a)
        assert a.nbytes == self.nbytes
        _ck(hip().hipMemcpy(self.ptr, a.ctypes.data, a.nbytes, 1), "hipMemcpy H2D")
        return self

    def copy_from(self, other):
        assert other.nbytes == self.nbytes
        _ck(hip().hipMemcpy(self.ptr, other.ptr, self.nbytes, 3), "hipMemcpy D2D")

    def free(self):
        if getattr(self, "ptr", None):
            hip().hipFree(self.ptr)
            self.ptr = None

    __del__ = free
