"""ctypes binding of libfdtd2d.so (include/fdtd2d.h).

This is the whole FFI: plain pointers and sizes, no torch types.  The library is
built in-tree by ``__graft_entry__.build()`` / ``fdtd-2d_amd/csrc/Makefile`` and is
required: there is no CPU fallback, a missing library or device raises.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# FDTD2D_ARITHMETIC=fused selects the tolerance build (libfdtd2d_fused.so: multiply-add pairs contracted into FMA,
# results within rounding of the reference's); default "exact" = value-identical to the reference.
# FDTD2D_LIB: alternative build of the same library (kernel A/B experiments only)
ARITHMETIC = os.environ.get("FDTD2D_ARITHMETIC", "exact")
if ARITHMETIC not in ("exact", "fused"):
    raise ImportError(f"FDTD2D_ARITHMETIC must be 'exact' or 'fused', not {ARITHMETIC!r}")
LIB_PATH = os.environ.get("FDTD2D_LIB") or os.path.join(
    HERE, "libfdtd2d.so" if ARITHMETIC == "exact" else "libfdtd2d_fused.so")

F32, F64 = 0, 1
BOUNDARY_NONE, BOUNDARY_MUR5, BOUNDARY_PML = 0, 1, 2
SRC_NONE, SRC_RICKER, SRC_SINUSOIDAL = 0, 1, 2
FIELD_EZ, FIELD_HX, FIELD_HY = 0, 1, 2

OPT_MAX_PASS_STEPS, OPT_BAND_ROWS, OPT_ZONE_SPLIT, OPT_LEVEL_SPLIT, OPT_SPLIT_WAVES, OPT_AUTOTUNE, OPT_XCD_MAP, OPT_SIDE_WAVES = 0, 1, 2, 3, 4, 5, 7, 8

E_ARG, E_NODEVICE, E_NOMEM, E_STATE, E_COURANT = -1, -2, -3, -4, -5

(INFO_ROWS, INFO_COLS, INFO_ROW0, INFO_NROWS, INFO_HALO, INFO_PITCH, INFO_DTYPE,
 INFO_BOUNDARY, INFO_DEVICE, INFO_EPS_UNIFORM, INFO_MU_UNIFORM, INFO_E_VALID_LO,
 INFO_E_VALID_HI, INFO_H_VALID_LO, INFO_H_VALID_HI, INFO_STEP, INFO_PASS_LAUNCHES,
 INFO_STEP_LAUNCHES, INFO_CYCLE_STEPS, INFO_LAST_BAND_ROWS, INFO_LAST_WAVES, INFO_LAST_EDGE_ROWS,
 INFO_LAST_PASS_STEPS, INFO_LAST_SIDE_WAVES, INFO_LAST_XCD_MAP) = range(25)

_vp, _i, _d, _ll = C.c_void_p, C.c_int, C.c_double, C.c_longlong

# name -> (restype, argtypes); every symbol include/fdtd2d.h declares
SIGNATURES = {
    "fdtd2d_create": (_i, [C.POINTER(_vp), _i, _i, _d, _d, _i, _i, _i]),
    "fdtd2d_create_slab": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _d, _d, _i, _i, _i]),
    "fdtd2d_destroy": (None, [_vp]),
    "fdtd2d_last_error": (C.c_char_p, [_vp]),
    "fdtd2d_info": (_ll, [_vp, _i]),
    "fdtd2d_set_stream": (_i, [_vp, _vp]),
    "fdtd2d_set_materials": (_i, [_vp, _vp, _vp, _i, C.POINTER(_d), _i]),
    "fdtd2d_set_materials_uniform": (_i, [_vp, _d, _d]),
    "fdtd2d_set_pml": (_i, [_vp, _vp, _vp, _i, _i]),
    "fdtd2d_transfer_ezx": (_i, [_vp, _vp, _i, _i]),
    "fdtd2d_courant": (_d, [_vp]),
    "fdtd2d_upload": (_i, [_vp, _vp, _vp, _vp, _i]),
    "fdtd2d_download": (_i, [_vp, _vp, _vp, _vp, _i]),
    "fdtd2d_reset": (_i, [_vp]),
    "fdtd2d_update_h": (_i, [_vp]),
    "fdtd2d_update_e": (_i, [_vp]),
    "fdtd2d_add_point": (_i, [_vp, _i, _i, _d]),
    "fdtd2d_set_source_extent": (_i, [_vp, _i, _i]),
    "fdtd2d_prepare": (_i, [_vp, _i]),
    "fdtd2d_prepare_run": (_i, [_vp, _i, _i, _i, _i]),
    "fdtd2d_set_probe": (_i, [_vp, _i, _i, C.c_longlong]),
    "fdtd2d_read_probe": (_i, [_vp, _vp, C.c_longlong, C.c_longlong]),
    "fdtd2d_run": (_i, [_vp, _i, _i, _i, C.POINTER(_d)]),
    "fdtd2d_pass_rows": (_i, [_vp, _i, _i, _i, _i, _i, C.POINTER(_d)]),
    "fdtd2d_pass_commit": (_i, [_vp]),
    "fdtd2d_run_waveform": (_i, [_vp, _i, _i, _i, _i, _d, _ll]),
    "fdtd2d_source_amplitude": (_d, [_i, _d, _d]),
    "fdtd2d_set_option": (_i, [_vp, _i, _ll]),
    "fdtd2d_set_shape": (_i, [_vp, _i, C.POINTER(_i), _i]),
    "fdtd2d_last_shape": (_i, [_vp, C.POINTER(_i), _i]),
    "fdtd2d_sync": (_i, [_vp]),
    "fdtd2d_halo_bytes": (_ll, [_vp]),
    "fdtd2d_halo_pack": (_i, [_vp, _i, _vp]),
    "fdtd2d_halo_unpack": (_i, [_vp, _i, _vp]),
    "fdtd2d_slab_attach": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fdtd2d_rccl_unique_id": (_i, [_vp]),
    "fdtd2d_rccl_selftest": (_i, [_i, _ll]),
    "fdtd2d_slab_attach_rccl": (_i, [_vp, _vp, _i, _i]),
    "fdtd2d_slab_detach": (_i, [_vp]),
    "fdtd2d_slab_ranks": (_ll, [_vp]),
    "fdtd2d_run_slab": (_i, [_vp, _i, _i, _i, _i, _i, C.POINTER(_d)]),
    "fdtd2d_set_dft": (_i, [_vp, _i, _i, _i, _i, _i, C.POINTER(_d), _i]),
    "fdtd2d_read_dft": (_i, [_vp, C.POINTER(_d), C.POINTER(_d)]),
    "fdtd2d_snapshot_index": (_i, [_vp, _d, _d, _i, _vp]),
    "fdtd2d_reduce": (_i, [_vp, _i, C.POINTER(_d), C.POINTER(_d)]),
    "fdtd2d_timer_start": (_i, [_vp]),
    "fdtd2d_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "fdtd2d_time_launches": (_i, [_vp, _i, _i, C.POINTER(C.c_float)]),
    "fdtd2d_clock_probe_start": (_i, [_vp, _i]),
    "fdtd2d_measure_copy": (_i, [_vp, _i, C.POINTER(_d)]),
    "fdtd2d_clock_probe_read": (_i, [_vp, C.POINTER(_d)]),
    "fdtd2d_bytes_per_cell_step": (_i, [_vp]),
    "fdtd2d_device_ptr": (_vp, [_vp, _i]),
    "fdtd2d_version": (C.c_char_p, []),
}

# transport callback of fdtd2d_slab_attach
EXCHANGE_FN = C.CFUNCTYPE(_i, _vp, _vp, _vp, _vp, _vp, _ll, _vp)

_lib = None


class Fdtd2dError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libfdtd2d error {code}: {msg}")
        self.code = code


def _share_torch_hip_runtime():
    """One HIP runtime per process.  ROCm wheels of torch bundle their own libamdhip64.so / libhsa-runtime64.so (same
    SONAMEs as /opt/rocm's, other files).  Imported first, torch's copies satisfy libfdtd2d.so's dependency and the process
    holds one runtime; imported AFTER libfdtd2d.so, torch maps its copies beside the system's, and the runtime that
    initialises second can find the GPU taken (round 3: "No HIP GPUs are available" in a test process after 440 tests).
    SlabRunner and bench.py need torch in the same process (streams, RCCL), so where torch is installed its copy is
    mapped first, whatever the import order.  FDTD2D_SYSTEM_HIP=1 keeps the system's runtime."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("FDTD2D_SYSTEM_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    for d in (spec.submodule_search_locations or []) if spec else []:
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = os.path.join(d, "lib", name)
            if os.path.exists(path):
                try:
                    C.CDLL(path, mode=C.RTLD_GLOBAL)
                except OSError:
                    return


def load():
    """Load libfdtd2d.so and declare every prototype.  Raises if it is not built."""
    global _lib
    if _lib is None:
        _share_torch_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C fdtd-2d_amd/csrc`). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(handle, rc: int):
    if rc != 0:
        msg = load().fdtd2d_last_error(handle)
        raise Fdtd2dError(rc, msg.decode() if msg else "")
    return rc
