"""Device-resident FDTD engine: the fast path of the drop-in.

``Engine`` owns one libfdtd2d handle (one GPU, whole grid or one row slab) and keeps
Ez/Hx/Hy in HBM across steps; host arrays only cross at upload/download.  It is a
thin object wrapper over the C ABI -- all arithmetic is in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi

_DT = {np.dtype(np.float32): _abi.F32, np.dtype(np.float64): _abi.F64}
_BOUNDARY = {"none": _abi.BOUNDARY_NONE, "mur": _abi.BOUNDARY_MUR5, "mur5": _abi.BOUNDARY_MUR5,
             "pml": _abi.BOUNDARY_PML}


def _code(dtype) -> int:
    try:
        return _DT[np.dtype(dtype)]
    except KeyError:
        raise TypeError(f"unsupported dtype {dtype}; use float32 or float64") from None


def _host(a, name):
    """Borrow a host array: C-contiguous float32/float64, no copy when already so."""
    a = np.asarray(a)
    if a.dtype not in _DT:
        a = a.astype(np.float64)
    if not a.flags.c_contiguous:
        a = np.ascontiguousarray(a)
    return a


class Engine:
    """One grid (or row slab) resident on one MI355X.

    rows, cols : global grid (reference shapes: Ez rows x cols, Hx rows x (cols-1),
                 Hy (rows-1) x cols, python-src/main.py:79-85)
    dt, dx     : step sizes (python-src/fdtd.py:16-17)
    dtype      : arithmetic/storage type on the device, float32 (default) or float64
    boundary   : "mur" (reference, main.py:29-61) or "none"
    slab       : None for the whole grid, or (row0, nrows, halo) for a row slab
    """

    def __init__(self, rows, cols, dt=5e-14, dx=1e-4, dtype=np.float32, boundary="mur",
                 device=0, slab=None):
        self._lib = _abi.load()
        self._h = C.c_void_p()
        self.rows, self.cols, self.dt, self.dx = int(rows), int(cols), float(dt), float(dx)
        self.dtype = np.dtype(dtype)
        self.boundary = boundary
        code, bcode = _code(dtype), _BOUNDARY[boundary]
        if slab is None:
            self.row0, self.nrows, self.halo = 0, self.rows, 0
            rc = self._lib.fdtd2d_create(C.byref(self._h), self.rows, self.cols, self.dt, self.dx,
                                         code, bcode, int(device))
        else:
            self.row0, self.nrows, self.halo = (int(v) for v in slab)
            rc = self._lib.fdtd2d_create_slab(C.byref(self._h), self.rows, self.cols, self.row0,
                                              self.nrows, self.halo, self.dt, self.dx, code, bcode,
                                              int(device))
        if rc != 0:
            msg = self._lib.fdtd2d_last_error(None).decode()
            self._h = C.c_void_p()
            raise _abi.Fdtd2dError(rc, msg)
        self.halo = int(self._lib.fdtd2d_info(self._h, _abi.INFO_HALO))

    # -- lifetime -------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.fdtd2d_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc):
        return _abi.check(self._h, rc)

    def info(self, what: int) -> int:
        return int(self._lib.fdtd2d_info(self._h, what))

    # -- rows this handle stores / owns ------------------------------------------------
    @property
    def stored_rows(self):
        return max(0, self.row0 - self.halo), min(self.rows, self.row0 + self.nrows + self.halo)

    @property
    def owned_rows(self):
        return self.row0, self.row0 + self.nrows

    # -- materials ------------------------------------------------------------------
    def set_materials(self, eps=None, mu=None, *, corner=None, allow_uniform=True):
        """eps, mu: arrays for the STORED rows (whole grid for a non-slab engine), or
        scalars, or None for vacuum (material_init(None, ...), main.py:100-106)."""
        from .api import EPS0, MU0
        eps = EPS0 if eps is None else eps
        mu = MU0 if mu is None else mu
        if np.isscalar(eps) and np.isscalar(mu):
            self._ck(self._lib.fdtd2d_set_materials_uniform(self._h, float(eps), float(mu)))
            return self
        lo, hi = self.stored_rows
        shape = (hi - lo, self.cols)
        e = _host(np.broadcast_to(eps, shape) if np.isscalar(eps) else eps, "eps")
        m = _host(np.broadcast_to(mu, shape) if np.isscalar(mu) else mu, "mu")
        if e.shape != shape or m.shape != shape:
            raise ValueError(f"eps/mu must have shape {shape} (stored rows {lo}..{hi}), "
                             f"got {e.shape} and {m.shape}")
        if e.dtype != m.dtype:
            m = m.astype(e.dtype)
        cptr = None
        if corner is not None:
            cptr = (C.c_double * 2)(float(corner[0]), float(corner[1]))
        self._ck(self._lib.fdtd2d_set_materials(self._h, e.ctypes.data, m.ctypes.data,
                                                _code(e.dtype), cptr, int(bool(allow_uniform))))
        return self

    def set_pml(self, L=40, m=3, R0=1e-6, courant00=None, profiles=None):
        """boundary="pml" only: build the split-field PML's factor arrays (pml_profiles) for
        the global grid and hand them to the engine.  courant00 = Courant number of the
        [0,0] material cell (default: vacuum).  profiles: the eight factor arrays themselves
        (keys ahr bhr aer ber of length rows, ahc bhc aec bec of length cols) instead of the
        graded ones; arrays that are not exactly 1 outside the L-cell layer run on the 8-step
        kernel (cycle_steps says so)."""
        from .api import pml_profiles, EPS0, MU0
        if courant00 is None:
            courant00 = (1 / np.sqrt(EPS0 * MU0) * self.dt) / self.dx
        P = profiles if profiles is not None else pml_profiles(self.rows, self.cols, courant00, L, m, R0, self.dtype)
        P = {k: np.asarray(P[k], dtype=self.dtype) for k in ("ahr", "bhr", "aer", "ber", "ahc", "bhc", "aec", "bec")}
        if any(P[k].shape != (self.rows,) for k in ("ahr", "bhr", "aer", "ber")) or \
                any(P[k].shape != (self.cols,) for k in ("ahc", "bhc", "aec", "bec")):
            raise ValueError("PML factor arrays must have length rows (..r) / cols (..c)")
        rowf = np.ascontiguousarray(np.concatenate([P["ahr"], P["bhr"], P["aer"], P["ber"]]))
        colf = np.ascontiguousarray(np.concatenate([P["ahc"], P["bhc"], P["aec"], P["bec"]]))
        self._ck(self._lib.fdtd2d_set_pml(self._h, rowf.ctypes.data, colf.ctypes.data, _code(self.dtype),
                                          int(L)))
        return self

    def upload_ezx(self, Ezx):
        a = np.ascontiguousarray(Ezx, dtype=self.dtype)
        if a.shape != (self.nrows, self.cols):
            raise ValueError(f"Ezx must have shape {(self.nrows, self.cols)}")
        self._ck(self._lib.fdtd2d_transfer_ezx(self._h, a.ctypes.data, _code(a.dtype), 1))
        return self

    def download_ezx(self):
        a = np.empty((self.nrows, self.cols), self.dtype)
        self._ck(self._lib.fdtd2d_transfer_ezx(self._h, a.ctypes.data, _code(a.dtype), 0))
        return a

    def courant(self) -> float:
        return float(self._lib.fdtd2d_courant(self._h))

    @property
    def bytes_per_cell_step(self) -> int:
        return int(self._lib.fdtd2d_bytes_per_cell_step(self._h))

    # -- field transfer -------------------------------------------------------------
    def _field_shapes(self):
        r0, r1 = self.owned_rows
        return ((self.nrows, self.cols), (self.nrows, self.cols - 1),
                (min(r1, self.rows - 1) - r0, self.cols))

    def upload(self, Ez=None, Hx=None, Hy=None):
        """Host -> device for the owned rows (reference shapes, any float dtype)."""
        arrs, code = [], None
        for a, shp, nm in zip((Ez, Hx, Hy), self._field_shapes(), ("Ez", "Hx", "Hy")):
            if a is None:
                arrs.append(None)
                continue
            a = _host(a, nm)
            if a.shape != shp:
                raise ValueError(f"{nm} must have shape {shp}, got {a.shape}")
            if code is None:
                code = _code(a.dtype)
            elif _code(a.dtype) != code:
                a = a.astype(np.float64 if code == _abi.F64 else np.float32)
            arrs.append(a)
        if code is None:
            return self
        ptr = [None if a is None else a.ctypes.data for a in arrs]
        self._ck(self._lib.fdtd2d_upload(self._h, ptr[0], ptr[1], ptr[2], code))
        return self

    def download(self, Ez=None, Hx=None, Hy=None, dtype=None):
        """Device -> host.  With no arguments returns new arrays (Ez, Hx, Hy) of the
        engine dtype (or `dtype`); given arrays are filled in place and returned."""
        shapes = self._field_shapes()
        given = [Ez, Hx, Hy]
        if all(a is None for a in given):
            dt_ = np.dtype(dtype or self.dtype)
            given = [np.empty(s, dt_) for s in shapes]
        code = None
        for a, shp, nm in zip(given, shapes, ("Ez", "Hx", "Hy")):
            if a is None:
                continue
            if not isinstance(a, np.ndarray) or not a.flags.c_contiguous or a.dtype not in _DT:
                raise TypeError(f"{nm} must be a C-contiguous float32/float64 ndarray")
            if a.shape != shp:
                raise ValueError(f"{nm} must have shape {shp}, got {a.shape}")
            c = _code(a.dtype)
            if code is not None and c != code:
                raise TypeError("output arrays must share one dtype")
            code = c
        ptr = [None if a is None else a.ctypes.data for a in given]
        self._ck(self._lib.fdtd2d_download(self._h, ptr[0], ptr[1], ptr[2], code))
        return tuple(given)

    def reset(self):
        self._ck(self._lib.fdtd2d_reset(self._h))
        return self

    # -- hot path -------------------------------------------------------------------
    def update_h(self):
        self._ck(self._lib.fdtd2d_update_h(self._h))

    def update_e(self):
        self._ck(self._lib.fdtd2d_update_e(self._h))

    def add_point(self, row, col, amp):
        self._ck(self._lib.fdtd2d_add_point(self._h, int(row), int(col), float(amp)))

    def prepare(self, nsteps, src_row=None, src_col=None):
        """Measure the launch shapes run(nsteps[, src_row, src_col, amps]) will use now (trial
        launches that leave the fields untouched) instead of inside its first passes."""
        if src_row is None:
            self._ck(self._lib.fdtd2d_prepare(self._h, int(nsteps)))
        else:
            self._ck(self._lib.fdtd2d_prepare_run(self._h, int(nsteps), int(src_row), int(src_col), 1))
        return self

    def set_probe(self, row, col, capacity):
        """Record Ez[row, col] after every step of the following run() calls (capacity samples,
        also the steps inside temporally blocked passes); capacity 0 removes the probe."""
        self._ck(self._lib.fdtd2d_set_probe(self._h, int(row), int(col), int(capacity)))
        self._probe = (int(capacity), self.step_count)
        return self

    def read_probe(self, first=0, count=None):
        """float64 samples [first, first + count) of the probe (waits for the stream)."""
        if count is None:
            cap, step0 = getattr(self, "_probe", (0, 0))
            count = max(0, min(cap, self.step_count - step0) - first)
        out = np.zeros(int(count), np.float64)
        self._ck(self._lib.fdtd2d_read_probe(self._h, out.ctypes.data, int(first), int(count)))
        return out

    def set_dft(self, window, omegas, every=16):
        """Running Fourier transform of Ez: window = (row0, col0, nrows, ncols); omegas = angular
        frequencies (at most 16); after every `every`-th step n the engine adds Ez * exp(-1j * omega * n * dt) per cell
        and frequency (float64, on the device).  every = 16 rides on the 16-step passes for free.  omegas = () removes it."""
        w = np.ascontiguousarray(omegas, dtype=np.float64).reshape(-1)
        r0, c0, nr, nc = (int(v) for v in window)
        self._ck(self._lib.fdtd2d_set_dft(self._h, r0, c0, nr, nc, int(w.size), w.ctypes.data_as(C.POINTER(C.c_double)), int(every)))
        lo, hi = max(r0, self.row0), min(r0 + nr, self.row0 + self.nrows)
        self._dft = (int(w.size), max(0, hi - lo), nc)
        return self

    def read_dft(self):
        """complex128 array (frequencies, owned window rows, window columns) of the accumulated transform."""
        nf, nr, nc = getattr(self, "_dft", (0, 0, 0))
        re, im = np.zeros((nf, nr, nc)), np.zeros((nf, nr, nc))
        if nf:
            self._ck(self._lib.fdtd2d_read_dft(self._h, re.ctypes.data_as(C.POINTER(C.c_double)),
                                               im.ctypes.data_as(C.POINTER(C.c_double))))
        return re + 1j * im

    def set_source_extent(self, nrows=1, ncols=1):
        """Line / patch sources: the source of add_point / run / pass_rows becomes the rectangle
        of nrows x ncols cells starting at the (row, col) given there (default one cell)."""
        self._ck(self._lib.fdtd2d_set_source_extent(self._h, int(nrows), int(ncols)))
        return self

    def run(self, nsteps, src_row=0, src_col=0, amps=None):
        """nsteps of H -> E -> source (python-src/fdtd.py:30-34), asynchronous.
        amps: float64 amplitude per step (None = no source)."""
        if amps is None:
            self._ck(self._lib.fdtd2d_run(self._h, int(nsteps), 0, 0, None))
            return self
        a = np.ascontiguousarray(amps, dtype=np.float64)
        if a.shape[0] < nsteps:
            raise ValueError("amps shorter than nsteps")
        self._ck(self._lib.fdtd2d_run(self._h, int(nsteps), int(src_row), int(src_col),
                                      a.ctypes.data_as(C.POINTER(C.c_double))))
        return self

    def pass_rows(self, nt, row_lo, row_hi, src_row=0, src_col=0, amps=None):
        """Issue output rows [row_lo,row_hi) of one nt-step pass on the current stream."""
        ap = None
        if amps is not None:
            a = np.ascontiguousarray(amps, dtype=np.float64)
            if a.shape[0] < nt:
                raise ValueError("amps shorter than the pass")
            ap = a.ctypes.data_as(C.POINTER(C.c_double))
        self._ck(self._lib.fdtd2d_pass_rows(self._h, int(nt), int(row_lo), int(row_hi), int(src_row),
                                            int(src_col), ap))

    def pass_commit(self):
        self._ck(self._lib.fdtd2d_pass_commit(self._h))

    def run_waveform(self, nsteps, kind="ricker", src_row=0, src_col=0, fc=30e9, step0=0):
        k = {"none": _abi.SRC_NONE, "ricker": _abi.SRC_RICKER,
             "sinusoidal": _abi.SRC_SINUSOIDAL}[kind]
        self._ck(self._lib.fdtd2d_run_waveform(self._h, int(nsteps), k, int(src_row),
                                               int(src_col), float(fc), int(step0)))
        return self

    def set_option(self, max_pass_steps=None, band_rows=None, zone_split=None, level_split=None,
                   split_waves=None, autotune=None, xcd_map=None, side_waves=None):
        """Speed knobs of run(): longest temporally blocked pass (0 = single-step kernels
        only) and rows per streaming band.  Results do not depend on them."""
        if max_pass_steps is not None:
            self._ck(self._lib.fdtd2d_set_option(self._h, _abi.OPT_MAX_PASS_STEPS, int(max_pass_steps)))
        if band_rows is not None:
            self._ck(self._lib.fdtd2d_set_option(self._h, _abi.OPT_BAND_ROWS, int(band_rows)))
        if zone_split is not None:
            self._ck(self._lib.fdtd2d_set_option(self._h, _abi.OPT_ZONE_SPLIT, int(zone_split)))
        if level_split is not None:
            self._ck(self._lib.fdtd2d_set_option(self._h, _abi.OPT_LEVEL_SPLIT, int(level_split)))
        if autotune is not None:
            self._ck(self._lib.fdtd2d_set_option(self._h, _abi.OPT_AUTOTUNE, int(bool(autotune))))
        if split_waves is not None:
            self._ck(self._lib.fdtd2d_set_option(self._h, _abi.OPT_SPLIT_WAVES, int(split_waves)))
        if side_waves is not None:
            self._ck(self._lib.fdtd2d_set_option(self._h, _abi.OPT_SIDE_WAVES, int(side_waves)))
        if xcd_map is not None:
            self._ck(self._lib.fdtd2d_set_option(self._h, _abi.OPT_XCD_MAP, int(xcd_map)))
        return self

    def set_shape(self, shape, pass_steps=0):
        """Launch shape of the passes of `pass_steps` steps (0 = the full-length ones): (band rows, waves per level
        group, edge band rows, waves side by side, xcd map, filler band rows, filler bands per strip, zone tiles of a
        float32 20-step pass fused 0 | 1) -- what
        last_shape returns, e.g. from another process.  Results do not depend on it."""
        v = [int(x) for x in shape]
        arr = (C.c_int * len(v))(*v)
        self._ck(self._lib.fdtd2d_set_shape(self._h, int(pass_steps), arr, len(v)))
        return self

    def sync(self):
        self._ck(self._lib.fdtd2d_sync(self._h))
        return self

    # -- halo / stream / timing -----------------------------------------------------------
    def set_stream(self, hip_stream: int | None):
        self._ck(self._lib.fdtd2d_set_stream(self._h, hip_stream))

    @property
    def halo_bytes(self) -> int:
        return int(self._lib.fdtd2d_halo_bytes(self._h))

    def halo_pack(self, side: int, dev_ptr: int):
        self._ck(self._lib.fdtd2d_halo_pack(self._h, int(side), dev_ptr))

    def halo_unpack(self, side: int, dev_ptr: int):
        self._ck(self._lib.fdtd2d_halo_unpack(self._h, int(side), dev_ptr))

    # -- the slab loop in C (fdtd2d_run_slab): no Python per exchange cycle ------------------------
    def slab_attach(self, bufs, fn):
        """bufs: {side: (send_ptr, recv_ptr)} device pointers of fdtd2d_halo_bytes each (side 0 = top
        neighbour, 1 = bottom); fn(send_top, recv_top, send_bottom, recv_bottom, nbytes, stream) is
        the transport (pointers are None for a side without neighbour), returning 0."""
        def cb(_ctx, st, rt, sb, rb, nbytes, stream):
            try:
                return int(fn(st, rt, sb, rb, int(nbytes), stream) or 0)
            except Exception:            # never unwind through the C frame
                import traceback
                traceback.print_exc()
                return 1
        self._exchange_cb = _abi.EXCHANGE_FN(cb)         # keep the thunk alive as long as the handle
        g = lambda side, k: bufs[side][k] if side in bufs else None
        self._ck(self._lib.fdtd2d_slab_attach(self._h, g(0, 0), g(0, 1), g(1, 0), g(1, 1),
                                              C.cast(self._exchange_cb, C.c_void_p), None))
        return self

    def slab_attach_rccl(self, unique_id: bytes, rank: int, world: int):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._ck(self._lib.fdtd2d_slab_attach_rccl(self._h, buf, int(rank), int(world)))
        return self

    @staticmethod
    def rccl_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        lib = _abi.load()
        rc = lib.fdtd2d_rccl_unique_id(buf)
        if rc:
            raise _abi.Fdtd2dError(rc, lib.fdtd2d_last_error(None).decode())
        return buf.raw

    @staticmethod
    def rccl_selftest(device: int = 0, count: int = 1 << 18) -> None:
        """One-rank RCCL communicator + a grouped ncclSend / ncclRecv to itself through the entry
        points the slab loop's built-in transport uses (runs on a single GPU); raises on failure."""
        lib = _abi.load()
        rc = lib.fdtd2d_rccl_selftest(int(device), int(count))
        if rc:
            raise _abi.Fdtd2dError(rc, lib.fdtd2d_last_error(None).decode())

    def slab_ranks(self):
        """(rank, ranks) of the RCCL communicator behind the built-in transport as RCCL reports them; None for a
        caller-supplied transport."""
        v = int(self._lib.fdtd2d_slab_ranks(self._h))
        if v < 0:
            self._ck(v)
        return (v >> 16, v & 0xffff) if v else None

    def run_slab(self, nsteps, cycle, overlap, src_row=0, src_col=0, amps=None):
        ap = None
        if amps is not None:
            a = np.ascontiguousarray(amps, dtype=np.float64)
            if a.shape[0] < nsteps:
                raise ValueError("amps shorter than nsteps")
            ap = a.ctypes.data_as(C.POINTER(C.c_double))
        self._ck(self._lib.fdtd2d_run_slab(self._h, int(nsteps), int(cycle), int(bool(overlap)), int(src_row),
                                           int(src_col), ap))
        return self

    # -- consumers of the fields next to the loop ---------------------------------------------
    def snapshot_index(self, vmin, vmax, stride=1):
        """uint8 colour-map indices of Ez (owned rows whose global index is a multiple of
        `stride`, every stride-th column), computed and decimated on the device."""
        r0, r1 = self.owned_rows
        first = -(-r0 // stride) * stride
        nro = (r1 - 1 - first) // stride + 1 if first < r1 else 0
        out = np.empty((nro, (self.cols - 1) // stride + 1), np.uint8)
        if nro:
            self._ck(self._lib.fdtd2d_snapshot_index(self._h, float(vmin), float(vmax), int(stride),
                                                     out.ctypes.data))
        return out

    def reduce(self, field="Ez"):
        """(sum of squares, max |.|) of a field over the owned rows, reduced on the device."""
        f = {"Ez": _abi.FIELD_EZ, "Hx": _abi.FIELD_HX, "Hy": _abi.FIELD_HY}[field]
        s, m = C.c_double(), C.c_double()
        self._ck(self._lib.fdtd2d_reduce(self._h, f, C.byref(s), C.byref(m)))
        return float(s.value), float(m.value)

    def measure_copy(self, reps=4) -> float:
        """GB/s (read + written) of a plain copy of the three field arrays on this device, now."""
        v = C.c_double()
        self._ck(self._lib.fdtd2d_measure_copy(self._h, int(reps), C.byref(v)))
        return float(v.value)

    def clock_probe_start(self, micros):
        """Start the shader-clock probe (runs for `micros` us beside whatever is launched next)."""
        self._ck(self._lib.fdtd2d_clock_probe_start(self._h, int(micros)))

    def clock_probe_read(self):
        """Shader clock in MHz per XCC id (0.0 where no probe workgroup landed)."""
        out = (C.c_double * 8)()
        self._ck(self._lib.fdtd2d_clock_probe_read(self._h, out))
        return [float(v) for v in out]

    def timer_start(self):
        self._ck(self._lib.fdtd2d_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._ck(self._lib.fdtd2d_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)

    def time_launches(self, nlaunch=32, steps_each=8):
        """Milliseconds of each of `nlaunch` back-to-back passes (own HIP event pair each)."""
        out = (C.c_float * int(nlaunch))()
        self._ck(self._lib.fdtd2d_time_launches(self._h, int(nlaunch), int(steps_each), out))
        return np.array(out[:], dtype=np.float64)

    @property
    def last_shape(self):
        """(band rows, waves per level group, band rows of the first / last strip, waves side by side, xcd map, filler
        band rows, filler bands per strip, zone tiles of a 20-step pass fused into the bulk launch) of the last pass."""
        out = (C.c_int * 8)()
        self._ck(self._lib.fdtd2d_last_shape(self._h, out, 8))
        return tuple(int(v) for v in out)

    @property
    def last_pass_steps(self) -> int:
        """Kernel length (1, 2, 4, 8, 16 or 20 steps) of the last temporally blocked pass; 0 before the first."""
        return self.info(_abi.INFO_LAST_PASS_STEPS)

    @property
    def cycle_steps(self) -> int:
        """Longest temporally blocked pass the current configuration runs (16 / 8 / 0)."""
        return self.info(_abi.INFO_CYCLE_STEPS)

    @property
    def step_count(self) -> int:
        return self.info(_abi.INFO_STEP)
