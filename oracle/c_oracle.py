"""ctypes loader for the C oracle (oracle/fdtd_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Same arrays and shapes as the reference (Ez RxC, Hx Rx(C-1), Hy (R-1)xC), all
C-contiguous NumPy arrays of one dtype (float32 or float64), updated in place.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfdtd_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "fdtd_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libfdtd_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_ricker.restype = C.c_double
        _lib.orc_ricker.argtypes = [C.c_double, C.c_double]
        _lib.orc_mur_coef_f32.restype = C.c_float
        _lib.orc_mur_coef_f32.argtypes = [C.c_float, C.c_float, C.c_double, C.c_double]
        _lib.orc_mur_coef_f64.restype = C.c_double
        _lib.orc_mur_coef_f64.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double]
    return _lib


def cpu_share() -> int:
    """CPUs this process may really use: the cgroup's CPU quota where there is one (a GPU box hands a job 16 of its 256
    hardware threads; 256 OpenMP threads on that share made a 512 x 512 oracle run take 73 s), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p_))
        except (OSError, ValueError):
            pass
    return n


def set_threads(rows, cols):
    """OpenMP threads for a grid: no more than the CPU share, no more than one per 32 Ki cells."""
    n = max(1, min(cpu_share(), rows * cols // 32768))
    lib().orc_set_threads(n)
    return n


def _suf(a):
    if a.dtype == np.float32:
        return "f32"
    if a.dtype == np.float64:
        return "f64"
    raise TypeError(f"unsupported dtype {a.dtype}")


def _p(a, dtype):
    if a.dtype != dtype or not a.flags.c_contiguous:
        raise TypeError("oracle arrays must share one dtype and be C-contiguous")
    return a.ctypes.data_as(C.c_void_p)


def update_h(Ez, Hx, Hy, mu, eps, dt, dx):
    R, Cc = Ez.shape
    f = getattr(lib(), "orc_update_h_" + _suf(Ez))
    rc = f(_p(Ez, Ez.dtype), _p(Hx, Ez.dtype), _p(Hy, Ez.dtype), _p(mu, Ez.dtype),
           C.c_int(R), C.c_int(Cc), C.c_double(dt), C.c_double(dx))
    if rc:
        raise RuntimeError(f"orc_update_h failed: {rc}")
    return Hx, Hy


def update_e(Ez, Hx, Hy, mu, eps, dt, dx):
    R, Cc = Ez.shape
    f = getattr(lib(), "orc_update_e_" + _suf(Ez))
    rc = f(_p(Ez, Ez.dtype), _p(Hx, Ez.dtype), _p(Hy, Ez.dtype), _p(mu, Ez.dtype),
           _p(eps, Ez.dtype), C.c_int(R), C.c_int(Cc), C.c_double(dt), C.c_double(dx),
           C.c_void_p(None))
    if rc:
        raise RuntimeError(f"orc_update_e failed: {rc}")
    return Ez


def add_point(Ez, row, col, amp):
    R, Cc = Ez.shape
    f = getattr(lib(), "orc_add_point_" + _suf(Ez))
    rc = f(_p(Ez, Ez.dtype), C.c_int(R), C.c_int(Cc), C.c_int(row), C.c_int(col),
           C.c_double(amp))
    if rc:
        raise RuntimeError(f"orc_add_point failed: {rc}")
    return Ez


def run(Ez, Hx, Hy, eps, mu, dt, dx, nsteps, src_row, src_col, amps=None, fc=30e9, step0=0):
    R, Cc = Ez.shape
    f = getattr(lib(), "orc_run_" + _suf(Ez))
    set_threads(R, Cc)
    if amps is not None:
        amps = np.ascontiguousarray(amps, dtype=np.float64)
        assert amps.shape[0] >= nsteps
        ap = amps.ctypes.data_as(C.c_void_p)
    else:
        ap = C.c_void_p(None)
    rc = f(_p(Ez, Ez.dtype), _p(Hx, Ez.dtype), _p(Hy, Ez.dtype), _p(eps, Ez.dtype),
           _p(mu, Ez.dtype), C.c_int(R), C.c_int(Cc), C.c_double(dt), C.c_double(dx),
           C.c_int(nsteps), C.c_int(src_row), C.c_int(src_col), ap, C.c_double(fc),
           C.c_longlong(step0))
    if rc:
        raise RuntimeError(f"orc_run failed: {rc}")
    return Ez, Hx, Hy


def ricker(t, fc):
    return lib().orc_ricker(t, fc)


def num_threads():
    return lib().orc_num_threads()
