"""Oracle-backed stand-in for the HIP Engine -- TEST INFRASTRUCTURE ONLY (CPU, gloo tests).

Implements the slice of the Engine interface that fdtd2d_amd.slab.SlabRunner uses, so that the
runner's sequencing (slab plan, who packs what, exchange order, source handling, validity) is
exercised without a GPU.  It keeps GLOBAL-size arrays whose rows outside the stored range
are zeros and advances them with the NumPy oracle: by the domain-of-dependence argument the
owned rows are exact after n <= halo steps if and only if the halo rows were filled with the
neighbours' true values -- which is exactly what the runner is responsible for.
"""
import ctypes

import os

import numpy as np

from oracle import fdtd_numpy as onp


def _view(ptr, n, dtype):
    ct = ctypes.c_float if np.dtype(dtype) == np.float32 else ctypes.c_double
    return np.ctypeslib.as_array((ct * n).from_address(ptr))


class FakeEngine:
    buffer_device = "cpu"
    cycle_steps = 8          # like the HIP engine with array materials: 8 steps per exchange

    def __init__(self, rows, cols, dt, dx, dtype=np.float32, boundary="mur", device=0, slab=None):
        self.cycle_steps = int(os.environ.get("FAKE_ENGINE_CYCLE", "8"))
        assert boundary == "mur"
        self.rows, self.cols, self.dt, self.dx, self.dtype = rows, cols, dt, dx, np.dtype(dtype)
        self.row0, self.nrows, self.halo = (0, rows, 0) if slab is None else slab
        self.Ez, self.Hx, self.Hy = onp.grid_zeros(rows, cols, self.dtype)
        self.eps = self.mu = None
        self.valid = [max(0, self.row0 - self.halo) if slab is None else self.row0,
                      self.row0 + self.nrows]

    @property
    def stored_rows(self):
        return max(0, self.row0 - self.halo), min(self.rows, self.row0 + self.nrows + self.halo)

    @property
    def owned_rows(self):
        return self.row0, self.row0 + self.nrows

    @property
    def halo_bytes(self):
        return 3 * self.halo * self.cols * self.dtype.itemsize

    def set_stream(self, s):
        pass

    def set_option(self, max_pass_steps=None, **_):
        if max_pass_steps is not None:
            self.cycle_steps = int(max_pass_steps)
        return self

    def sync(self):
        pass

    def close(self):
        pass

    def set_materials(self, eps=None, mu=None, *, corner=None, allow_uniform=True):
        lo, hi = self.stored_rows
        E = np.full((self.rows, self.cols), onp.EPS0, self.dtype)
        M = np.full((self.rows, self.cols), onp.MU0, self.dtype)
        if np.isscalar(eps):
            E[:] = eps
            M[:] = mu
        else:
            E[lo:hi] = eps
            M[lo:hi] = mu
            if lo > 0:            # the Mur factor reads the global [0,0] cell (main.py:30)
                E[0, 0], M[0, 0] = corner
        self.eps, self.mu = E, M

    def upload(self, Ez=None, Hx=None, Hy=None):
        r0, r1 = self.owned_rows
        if Ez is not None:
            self.Ez[r0:r1] = Ez
        if Hx is not None:
            self.Hx[r0:r1] = Hx
        if Hy is not None:
            self.Hy[r0:min(r1, self.rows - 1)] = Hy

    def download(self):
        r0, r1 = self.owned_rows
        assert self.valid[0] <= r0 and self.valid[1] >= r1, "owned rows not current"
        return (self.Ez[r0:r1].copy(), self.Hx[r0:r1].copy(),
                self.Hy[r0:min(r1, self.rows - 1)].copy())

    def _rows(self, side, pack):
        r0, r1 = self.owned_rows
        if pack:
            return (r0, r0 + self.halo) if side == 0 else (r1 - self.halo, r1)
        return (r0 - self.halo, r0) if side == 0 else (r1, r1 + self.halo)

    def halo_pack(self, side, ptr):
        a, b = self._rows(side, True)
        n = self.halo * self.cols
        buf = _view(ptr, 3 * n, self.dtype)
        if self.pending is not None:
            assert any(lo <= a and hi >= b for lo, hi in self.pending["rows"]), "edge rows not issued"
            cur = (self.Ez, self.Hx, self.Hy)
            self.Ez, self.Hx, self.Hy = self.pending["new"]
            try:
                self.pending, keep = None, self.pending
                self.halo_pack(side, ptr)
            finally:
                self.pending = keep
                self.Ez, self.Hx, self.Hy = cur
            return
        # Hx has cols-1 columns and Hy rows-1 rows in the reference layout; the message
        # carries `cols` per row like the device engine (missing entries are zeros)
        hx = np.zeros((self.halo, self.cols), self.dtype)
        hx[:, :-1] = self.Hx[a:b]
        hy = np.zeros((self.halo, self.cols), self.dtype)
        hy[:max(0, min(b, self.rows - 1) - a)] = self.Hy[a:min(b, self.rows - 1)]
        buf[:n] = self.Ez[a:b].ravel()
        buf[n:2 * n] = hx.ravel()
        buf[2 * n:] = hy.ravel()

    def halo_unpack(self, side, ptr):
        a, b = self._rows(side, False)
        n = self.halo * self.cols
        buf = _view(ptr, 3 * n, self.dtype)
        self.Ez[a:b] = buf[:n].reshape(self.halo, self.cols)
        self.Hx[a:b] = buf[n:2 * n].reshape(self.halo, self.cols)[:, :-1]
        hb = min(b, self.rows - 1)
        self.Hy[a:hb] = buf[2 * n:].reshape(self.halo, self.cols)[:hb - a]
        if side == 0:
            self.valid[0] = a
        else:
            self.valid[1] = b

    # -- a pass issued in pieces (Engine.pass_rows / pass_commit) ---------------------------
    pending = None

    def pass_rows(self, nt, row_lo, row_hi, src_row=0, src_col=0, amps=None):
        if self.pending is None:
            keep = (self.Ez, self.Hx, self.Hy, list(self.valid))
            self.Ez, self.Hx, self.Hy = self.Ez.copy(), self.Hx.copy(), self.Hy.copy()
            self.run(nt, src_row, src_col, amps)
            new = (self.Ez, self.Hx, self.Hy)
            self.Ez, self.Hx, self.Hy, self.valid = keep
            self.pending = dict(new=new, rows=[], nt=nt, amps=None if amps is None else np.array(amps[:nt]))
        p = self.pending
        assert p["nt"] == nt and (amps is None) == (p["amps"] is None)
        assert amps is None or np.array_equal(p["amps"], np.asarray(amps[:nt])), "pieces disagree on amps"
        p["rows"].append((row_lo, row_hi))

    def pass_commit(self):
        p = self.pending
        assert p is not None, "no pending pass"
        at = self.row0
        for lo, hi in sorted(p["rows"]):
            assert lo <= at, f"rows [{at},{lo}) of the pending pass were never issued"
            at = max(at, hi)
        assert at >= self.row0 + self.nrows
        self.Ez, self.Hx, self.Hy = p["new"]
        self.valid = [self.row0, self.row0 + self.nrows]
        self.pending = None

    def set_source_extent(self, nrows=1, ncols=1):
        self.extent = (int(nrows), int(ncols))
        return self

    def run(self, nsteps, src_row=0, src_col=0, amps=None):
        lo = 0 if self.row0 == 0 else self.valid[0]
        hi = self.rows if self.row0 + self.nrows == self.rows else self.valid[1]
        need_lo = 0 if self.row0 == 0 else self.row0 - nsteps
        need_hi = self.rows if hi == self.rows else self.row0 + self.nrows + nsteps
        assert lo <= need_lo and hi >= need_hi, f"stale halo: have [{lo},{hi}) need [{need_lo},{need_hi})"
        a = np.zeros(nsteps) if amps is None else np.asarray(amps, dtype=np.float64)
        for n in range(nsteps):
            onp.update_h(self.Ez, self.Hx, self.Hy, self.mu, self.eps, self.dt, self.dx)
            onp.update_e(self.Ez, self.Hx, self.Hy, self.mu, self.eps, self.dt, self.dx)
            if amps is not None:
                slo, shi = self.stored_rows
                nr, nc = getattr(self, "extent", (1, 1))
                r0, r1 = max(src_row, slo), min(src_row + nr, shi)
                if r0 < r1:
                    onp.add_source(self.Ez, r0, src_col, a[n], (r1 - r0, nc))
        self.valid = [self.row0, self.row0 + self.nrows]
