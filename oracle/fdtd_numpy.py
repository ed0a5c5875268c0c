"""CPU oracle (NumPy) for the 2D TE-mode FDTD leapfrog -- TEST INFRASTRUCTURE ONLY.

This file is a *restatement* of the reference algorithm, written from the
per-cell formulas, used as the checker for the HIP path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package (``fdtd-2d_amd/``) never does.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
(``/root/reference/python-src/main.py``) in the build container and stores its
outputs under ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks
every function below against those vectors bit-for-bit (float64 and float32
array runs).

Reference map (paths relative to the reference checkout):
  update_h          <- python-src/main.py:66-76   (update_Hx_Hy)
  update_e          <- python-src/main.py:12-63   (update_Ez: interior curl,
                                                   5-px Mur band, corner rule)
  mur_coefficient   <- python-src/main.py:30-31
  ricker_amplitude  <- python-src/main.py:182-187
  sinusoidal_amplitude <- python-src/main.py:190-195
  grid_zeros        <- python-src/main.py:79-85   (grid_init)
  vacuum_materials  <- python-src/main.py:100-106 (material_init(None, ...))
  courant_number    <- python-src/fdtd.py:25-28
  leapfrog          <- python-src/fdtd.py:30-34   (H -> E -> source, t = i*dt)

Index convention (as in the reference): first axis = row i, second = column j.
Ez is (R, C); Hx is (R, C-1); Hy is (R-1, C).

dtype semantics.  The arithmetic is written so that NumPy's own promotion
rules give the reference's result for whatever array dtype is passed in: with
float64 arrays everything is float64 (the reference default); with float32
arrays the Python-float scalars dt, dx are weak and every array expression
stays float32 -- exactly what the reference computes when handed float32
arrays.  The source amplitude is always a float64 scalar and is added to the
Ez cell in float64, then rounded to Ez's dtype (``Ez += float64 array`` in the
reference, fdtd.py:34).
"""
from __future__ import annotations

import numpy as np

EPS0 = 8.85418e-12            # main.py:100 (truncated literal, kept on purpose)
MU0 = 4 * np.pi * 1e-7        # main.py:101
BAND = 5                      # Mur band depth, main.py:34,38,44,48
MIN_GRID = 2 * BAND + 1       # below 11 the opposite bands overlap


def grid_zeros(rows: int, cols: int, dtype=np.float64):
    """Field shapes of main.py:79-85."""
    return (np.zeros((rows, cols), dtype),
            np.zeros((rows, cols - 1), dtype),
            np.zeros((rows - 1, cols), dtype))


def vacuum_materials(rows: int, cols: int, dtype=np.float64):
    """material_init(None, rows, cols), main.py:100-106. Returns (eps, mu)."""
    eps = (np.ones((rows, cols)) * EPS0).astype(dtype, copy=False)
    mu = (np.ones((rows, cols)) * MU0).astype(dtype, copy=False)
    return eps, mu


def courant_number(eps, mu, dt, dx):
    """fdtd.py:25-26: c*dt/dx with c from the smallest eps and mu."""
    c = 1 / np.sqrt(eps.min() * mu.min())
    return (c * dt) / dx


def mur_coefficient(mu00, eps00, dt, dx):
    """main.py:30-31: first-order Mur factor from the [0,0] material cell only."""
    c = 1 / np.sqrt(mu00 * eps00)
    return (c * dt - dx) / (c * dt + dx)


def ricker_amplitude(t, fc):
    """main.py:183-184 (scalar part of ricker())."""
    tau = np.pi * fc * (t - 1 / fc)
    return (1 - 2 * tau ** 2) * np.exp(-(tau ** 2))


def sinusoidal_amplitude(t, fc):
    """main.py:193-194 (scalar part of sinusoidal())."""
    envelope = 1 - np.exp(-((t - 3000 / fc) ** 2) / (2 * (2 / fc) ** 2))
    return envelope * np.sin(2 * np.pi * fc * t)


def h_coefficient(mu, dt, dx):
    """dt/(mu*dx) on the (R-1, C-1) block the H update touches (main.py:70,74)."""
    return dt / (mu[:-1, :-1] * dx)


def e_coefficient(eps, dt, dx):
    """dt/(eps*dx) on the interior block (main.py:27)."""
    return dt / (eps[1:-1, 1:-1] * dx)


def update_h(Ez, Hx, Hy, mu, eps, dt, dx):
    """H half-step, in place (main.py:66-76).  eps is unused, as in the reference."""
    ch = h_coefficient(mu, dt, dx)
    core = Ez[:-1, :-1]
    Hx[:-1, :] -= ch * (Ez[1:, :-1] - core)      # d/d(row)
    Hy[:, :-1] += ch * (Ez[:-1, 1:] - core)      # d/d(col)
    return Hx, Hy


def update_e(Ez, Hx, Hy, mu, eps, dt, dx):
    """E half-step, in place, as four barrier-separated stages (main.py:12-63).

    P = Ez before the call.  Stage A: interior curl update.  Stage B: left/right
    bands, rows 1..R-2.  Stage C: top/bottom bands, columns 1..C-2 (reads stage
    B values where the bands cross).  Stage D: the four 5x5 corner blocks (reads
    stage C values).  For grids >= 11x11 this equals the reference's sequential
    loops exactly: every loop iteration there reads only cells that a later
    iteration of the same loop (or nobody) writes.
    """
    R, C = Ez.shape
    if R < MIN_GRID or C < MIN_GRID:
        raise ValueError(f"grid {R}x{C} is below the {MIN_GRID}x{MIN_GRID} minimum "
                         "for the staged boundary form")
    P = Ez.copy()
    b = BAND

    # A
    curl = (Hy[1:, 1:-1] - Hy[1:, :-2]) - (Hx[1:-1, 1:] - Hx[:-2, 1:])
    Ez[1:-1, 1:-1] += curl * e_coefficient(eps, dt, dx)

    k = mur_coefficient(mu[0, 0], eps[0, 0], dt, dx)

    # B (both right-hand sides are evaluated before the band is overwritten)
    left = P[1:-1, 1:b + 1] + k * (Ez[1:-1, 1:b + 1] - P[1:-1, 0:b])
    right = P[1:-1, -b - 1:-1] + k * (Ez[1:-1, -b - 1:-1] - P[1:-1, -b:])
    Ez[1:-1, 0:b] = left
    Ez[1:-1, -b:] = right

    # C
    top = P[1:b + 1, 1:-1] + k * (Ez[1:b + 1, 1:-1] - P[0:b, 1:-1])
    bot = P[-b - 1:-1, 1:-1] + k * (Ez[-b - 1:-1, 1:-1] - P[-b:, 1:-1])
    Ez[0:b, 1:-1] = top
    Ez[-b:, 1:-1] = bot

    # D
    tl = (Ez[0:b, 1:b + 1] + Ez[1:b + 1, 0:b]) / 2
    tr = (Ez[0:b, -b - 1:-1] + Ez[1:b + 1, -b:]) / 2
    bl = (Ez[-b - 1:-1, 0:b] + Ez[-b:, 1:b + 1]) / 2
    br = (Ez[-b - 1:-1, -b:] + Ez[-b:, -b - 1:-1]) / 2
    Ez[0:b, 0:b] = tl
    Ez[0:b, -b:] = tr
    Ez[-b:, 0:b] = bl
    Ez[-b:, -b:] = br
    return Ez


def add_point(Ez, row, col, amp):
    """Ez += dense source with one non-zero cell (fdtd.py:34 with main.py:185-186).

    The reference adds a float64 array to Ez in place, so the one non-zero cell
    becomes round_to_Ez_dtype(float64(Ez[row,col]) + amp); all other cells get
    +0.0, which leaves their value unchanged.
    """
    Ez[row, col] = Ez.dtype.type(np.float64(Ez[row, col]) + np.float64(amp))
    return Ez


def add_source(Ez, row, col, amp, extent=(1, 1)):
    """`Ez += s` for a dense s that is `amp` on the rectangle of `extent` = (rows, cols) cells
    starting at (row, col) and 0 elsewhere: the line / patch source of SURVEY.md 8(f) N3.  The
    reference only builds the one-cell s (main.py:185-186); with more cells the in-place add
    rounds each of them exactly as it rounds that one."""
    nr, nc = extent
    blk = Ez[row:row + nr, col:col + nc]
    assert blk.shape == (nr, nc), "source rectangle outside the grid"
    blk[...] = (blk.astype(np.float64) + np.float64(amp)).astype(Ez.dtype)
    return Ez


def leapfrog(Ez, Hx, Hy, eps, mu, dt, dx, nsteps, src_row, src_col, amps=None,
             fc=30e9, step0=0, on_step=None, extent=(1, 1)):
    """The loop of fdtd.py:30-34: H, then E (A-D), then the point source.

    ``amps`` (float64, one per step) overrides the ricker waveform so that a
    caller can feed the oracle and the device path the very same numbers.
    """
    for n in range(nsteps):
        i = step0 + n
        update_h(Ez, Hx, Hy, mu, eps, dt, dx)
        update_e(Ez, Hx, Hy, mu, eps, dt, dx)
        a = amps[n] if amps is not None else ricker_amplitude(i * dt, fc)
        if tuple(extent) == (1, 1):
            add_point(Ez, src_row, src_col, a)
        else:
            add_source(Ez, src_row, src_col, a, extent)
        if on_step is not None:
            on_step(i, Ez, Hx, Hy)
    return Ez, Hx, Hy


def ring_resonator_eps(rows: int, cols: int, core_eps_r: float = 10.0, dtype=np.float64):
    """Synthetic permittivity map for BASELINE config 3.

    Background EPS0; a straight bus waveguide in rows [0.18R, 0.22R); a ring of
    mean radius 0.30R and width 0.04R centred at (0.54R, 0.50C); core relative
    permittivity ``core_eps_r`` (the reference's default black_point,
    main.py:89).  The geometry follows the recipe of region_drawer.py:13-28 /
    assets/ring_resonator.png; it is generated analytically (no image file).
    """
    i = np.arange(rows, dtype=np.float64)[:, None]
    j = np.arange(cols, dtype=np.float64)[None, :]
    core = np.zeros((rows, cols), dtype=bool)
    core |= (i >= np.floor(0.18 * rows)) & (i < np.floor(0.22 * rows))
    rad = np.sqrt((i - 0.54 * rows) ** 2 + (j - 0.50 * cols) ** 2)
    core |= np.abs(rad - 0.30 * rows) <= 0.02 * rows
    eps = np.where(core, core_eps_r * EPS0, EPS0)
    return eps.astype(dtype, copy=False)
