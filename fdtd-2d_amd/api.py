"""Drop-in call surface of the reference's FDTD path, backed by the HIP engine.

Same names, argument order and in-place semantics as python-src/main.py, plus the
`step` / `run_fdtd` wrappers BASELINE.json's north_star names (they are the loop of
python-src/fdtd.py:13-40, which the reference keeps under ``if __name__ == ...``).

Every function here computes on the GPU through libfdtd2d (ctypes, C ABI).  There is
no CPU fallback: without the built library or a gfx950 device these raise.

Host-side scalars (source waveforms, Courant number) are float64 NumPy, as in the
reference; only they are evaluated on the host.
"""
from __future__ import annotations

import numpy as np

from .engine import Engine

EPS0 = 8.85418e-12          # python-src/main.py:100 (truncated literal, kept)
MU0 = 4 * np.pi * 1e-7      # python-src/main.py:101


# ---- setup helpers (python-src/main.py:79-123) -------------------------------------------

def grid_init(rows: int, cols: int, dtype=np.float64):
    """Ez (rows, cols), Hx (rows, cols-1), Hy (rows-1, cols), zeros (main.py:79-85)."""
    return (np.zeros((rows, cols), dtype), np.zeros((rows, cols - 1), dtype),
            np.zeros((rows - 1, cols), dtype))


def material_init(path, rows: int, cols: int, black_point: float = 10.0):
    """(eps, mu) float64 arrays (main.py:88-123).  path=None -> vacuum; otherwise a
    grayscale image: black -> black_point*EPS0, white -> EPS0, LANCZOS-resized."""
    if path is None:
        return np.ones((rows, cols)) * EPS0, np.ones((rows, cols)) * MU0
    from PIL import Image
    img = Image.open(path).convert("L").resize((cols, rows), Image.LANCZOS)
    darkness = 1.0 - np.array(img, dtype=float) / 255.0
    eps = (1 + (black_point - 1) * darkness) * EPS0
    return eps, np.ones((rows, cols)) * MU0


def ricker_amplitude(t, fc):
    """Scalar waveform of main.py:183-184."""
    tau = np.pi * fc * (t - 1 / fc)
    return (1 - 2 * tau ** 2) * np.exp(-(tau ** 2))


def sinusoidal_amplitude(t, fc):
    """Scalar waveform of main.py:193-194."""
    envelope = 1 - np.exp(-((t - 3000 / fc) ** 2) / (2 * (2 / fc) ** 2))
    return envelope * np.sin(2 * np.pi * fc * t)


def ricker(rows, cols, x_pos, y_pos, t, fc):
    """Dense source array with one non-zero cell (main.py:182-187); x_pos = row."""
    src = np.zeros((rows, cols), dtype=float)
    src[x_pos, y_pos] = ricker_amplitude(t, fc)
    return src


def sinusoidal(rows, cols, x_pos, y_pos, t, fc):
    """main.py:190-195."""
    src = np.zeros((rows, cols), dtype=float)
    src[x_pos, y_pos] = sinusoidal_amplitude(t, fc)
    return src


# ---- absorbing layer for boundary="pml" (build-defined; grading as python-src/fdfd.py:16-30) -----

def pml_profiles(rows, cols, courant00, L=40, m=3, R0=1e-6, dtype=np.float64):
    """Loss factors a = (1-s)/(1+s), b = 1/(1+s) of the split-field PML, as eight 1-D arrays:
    rows {ahr, bhr (Hx, half positions), aer, ber (Ez)}, columns {ahc, bhc (Hy), aec, bec}.
    s(d) = s_max (d/L)^m with s_max = (m+1) ln(1/R0) courant00 / (4 L), d = depth in cells;
    L = 40 and m = 3 are the defaults of the reference's frequency-domain PML."""
    smax = (m + 1) * np.log(1.0 / R0) * courant00 / (4.0 * L)
    out = {}
    for tag, n in (("r", rows), ("c", cols)):
        pos = np.arange(n, dtype=np.float64)
        for kind, p in (("e", pos), ("h", pos + 0.5)):
            d = np.maximum(0.0, np.maximum(L - p, p - (n - 1 - L)))
            s = smax * (d / L) ** m
            out["a" + kind + tag] = ((1 - s) / (1 + s)).astype(dtype)
            out["b" + kind + tag] = (1 / (1 + s)).astype(dtype)
    return out


# ---- snapshots (python-src/main.py:153-179) ----------------------------------------------------

def snapshot_indices(Ez, vmax=20, vmin=-20):
    """LUT index matplotlib derives for `cmap((clip(Ez) - vmin)/(vmax - vmin))`, in Ez's dtype
    (main.py:155,167-168).  Host twin of Engine.snapshot_index()."""
    x = (np.clip(Ez, vmin, vmax) - vmin) / (vmax - vmin)
    y = x * 256
    idx = y.astype(int)
    idx[y == 256] = 255
    return np.clip(idx, 0, 255).astype(np.uint8)


def eps_background(eps):
    """uint8 gray level behind the field plot (main.py:157-165)."""
    eps = np.asarray(eps)
    eps_min = EPS0
    eps_max = np.max(eps)
    if eps_max == eps_min:
        return np.full(eps.shape, 255, dtype=np.uint8)
    frac = (eps - eps_min) / (eps_max - eps_min)
    return ((1 - frac) * 127 + 128).astype(np.uint8)


_snap_table = None


def snapshot_table():
    """(256, 256, 3) uint8: final pixel for (colour index, background gray) -- the seismic
    colour map at alpha 0.7 over the gray level (main.py:167-174), tabulated once."""
    global _snap_table
    if _snap_table is None:
        import matplotlib
        lut = matplotlib.colormaps["seismic"](np.arange(256))[:, :3]            # (256, 3) float64
        gray = np.arange(256, dtype=np.float64)[None, :, None] / 255
        alpha = 0.7
        _snap_table = ((lut[:, None, :] * alpha + gray * (1 - alpha)) * 255).astype(np.uint8)
    return _snap_table


def render_snapshot(indices, eps_gray):
    """RGB image (uint8) from colour indices and the eps background."""
    return snapshot_table()[indices, eps_gray]


def capture_snapshot(Ez, eps, path, vmax=20, vmin=-20):
    """Drop-in for main.py:153-179: seismic colour map of clip(Ez) at alpha 0.7 over the eps
    background, written as PNG.  Ez may be a host array or an Engine (then the colour index is
    computed on the device and only one byte per cell is downloaded)."""
    from PIL import Image
    idx = Ez.snapshot_index(vmin, vmax, 1) if isinstance(Ez, Engine) else snapshot_indices(np.asarray(Ez), vmax, vmin)
    img = render_snapshot(idx, eps_background(eps))
    if path is not None:
        Image.fromarray(img).save(path)
    return img


def courant_number(eps, mu, dt, dx):
    """fdtd.py:25-26."""
    c = 1 / np.sqrt(np.min(eps) * np.min(mu))
    return (c * dt) / dx


# ---- per-call drop-ins: upload -> kernel -> download (parity path, not the fast path) ----

_cache: dict = {}


def invalidate_cache():
    """Drop the engines the per-call drop-ins keep between calls (frees their device memory)."""
    while _cache:
        _cache.popitem()[1][0].close()


def _content_sig(a):
    """Content hash of a material array: any in-place edit changes it (a sum would not notice a
    moved structure)."""
    import zlib
    a = np.ascontiguousarray(a)
    return a.shape, a.dtype.str, zlib.adler32(a.view(np.uint8).reshape(-1))


def _engine_for(Ez, mu, eps, dt, dx) -> Engine:
    """One cached engine per (shape, dtype, dt, dx); materials are re-sent whenever the
    arrays' content differs from what the engine holds (content hash, not identity)."""
    Ez = np.asarray(Ez)
    if Ez.dtype not in (np.float32, np.float64):
        raise TypeError("fields must be float32 or float64 arrays")
    key = (Ez.shape, Ez.dtype.str, float(dt), float(dx))
    ent = _cache.get(key)
    if ent is None:
        if len(_cache) >= 4:
            _cache.pop(next(iter(_cache)))[0].close()
        ent = [Engine(Ez.shape[0], Ez.shape[1], dt, dx, dtype=Ez.dtype), None]
        _cache[key] = ent
    eng = ent[0]
    eps_a, mu_a = np.asarray(eps), np.asarray(mu)
    sig = (_content_sig(eps_a), _content_sig(mu_a))
    if ent[1] != sig:
        eng.set_materials(eps_a.astype(Ez.dtype, copy=False), mu_a.astype(Ez.dtype, copy=False))
        ent[1] = sig
    return eng


def _inplace(a, name):
    if not isinstance(a, np.ndarray) or not a.flags.c_contiguous or not a.flags.writeable:
        raise TypeError(f"{name} must be a writable C-contiguous ndarray (it is updated in place)")
    return a


def update_Hx_Hy(Ez, Hx, Hy, mu, eps, dt, dx):
    """H half-step on the GPU; mutates Hx, Hy in place and returns them (main.py:66-76)."""
    _inplace(Hx, "Hx"), _inplace(Hy, "Hy")
    eng = _engine_for(Ez, mu, eps, dt, dx)
    eng.upload(Ez, Hx, Hy)
    eng.update_h()
    eng.download(None, Hx, Hy)
    return Hx, Hy


def update_Ez(Ez, Hx, Hy, mu, eps, dt, dx):
    """E half-step (curl + 5-px Mur + corners) on the GPU; mutates and returns Ez
    (main.py:12-63)."""
    _inplace(Ez, "Ez")
    eng = _engine_for(Ez, mu, eps, dt, dx)
    eng.upload(Ez, Hx, Hy)
    eng.update_e()
    eng.download(Ez, None, None)
    return Ez


# ---- north-star wrappers: the loop of python-src/fdtd.py:30-34 ----------------------------

def _source_amp(source, t, rows, cols):
    """-> (row, col, amp) or (None, None, dense array) for the ways a source can be given."""
    if source is None:
        return None, None, None
    if callable(source):
        source = source(t)
    if isinstance(source, tuple) and len(source) == 4 and isinstance(source[0], str):
        kind, r, c, fc = source
        f = {"ricker": ricker_amplitude, "sinusoidal": sinusoidal_amplitude}[kind]
        return int(r), int(c), float(f(t, fc))
    arr = np.asarray(source)
    if arr.shape != (rows, cols):
        raise ValueError("dense source must have the shape of Ez")
    nz = np.flatnonzero(arr)
    if nz.size == 0:
        return None, None, None
    if nz.size == 1:
        r, c = divmod(int(nz[0]), cols)
        return r, c, float(arr[r, c])
    return None, None, arr


def step(E, Hx, Hy, eps, mu, source, t, *, dt=5e-14, dx=1e-4):
    """One leapfrog step on the GPU, in place: H, then E, then `E += source(t)`
    (fdtd.py:31-34).  source: None, a callable t -> array, a dense array, or a point
    spec ("ricker"|"sinusoidal", row, col, fc).  Returns (E, Hx, Hy)."""
    _inplace(E, "E"), _inplace(Hx, "Hx"), _inplace(Hy, "Hy")
    eng = _engine_for(E, mu, eps, dt, dx)
    eng.upload(E, Hx, Hy)
    eng.update_h()
    eng.update_e()
    r, c, amp = _source_amp(source, t, *E.shape)
    if r is not None:
        eng.add_point(r, c, amp)
    eng.download(E, Hx, Hy)
    if r is None and amp is not None:      # general dense source: host add, as the reference
        E += amp
    return E, Hx, Hy


def run_fdtd(rows=200, cols=200, dt=5e-14, dx=1e-4, nsteps=1000, eps=None, mu=None,
             source=("ricker", None, None, 30e9), boundary="mur", dtype=np.float64,
             devices=1, on_frame=None, nframes=200, device=0):
    """python-src/fdtd.py:13-40 without the video: zero fields, Courant check, nsteps of
    H -> E -> source with t = i*dt, fields resident on the GPU throughout.

    eps/mu: None (vacuum), scalars or (rows, cols) arrays.  source: (kind, row, col, fc)
    with row/col None = grid centre (fdtd.py:34), or None; an optional fifth element
    (nrows, ncols) makes it a line / patch source starting at (row, col).  on_frame(i, Ez) is called
    every nsteps//nframes steps with a host copy of Ez (the snapshot cadence of
    fdtd.py:36-38).  Returns (Ez, Hx, Hy) as host arrays of `dtype`.  dtype defaults to float64, what
    the reference computes in (main.py:81-85); BASELINE's GPU configurations pass float32.
    """
    if devices != 1:
        from .slab import run_fdtd_distributed
        return run_fdtd_distributed(rows, cols, dt, dx, nsteps, eps, mu, source, boundary,
                                    dtype, on_frame, nframes)
    eps_h = EPS0 if eps is None else eps
    mu_h = MU0 if mu is None else mu
    courant = courant_number(eps_h, mu_h, dt, dx)
    assert courant <= 1.0, f"Courant stability condition not met: {courant} > 1.0"
    with Engine(rows, cols, dt, dx, dtype=dtype, boundary=boundary, device=device) as eng:
        eng.set_materials(eps_h, mu_h)
        if boundary == "pml":
            e00 = eps_h if np.isscalar(eps_h) else np.asarray(eps_h)[0, 0]
            m00 = mu_h if np.isscalar(mu_h) else np.asarray(mu_h)[0, 0]
            eng.set_pml(courant00=(1 / np.sqrt(float(e00) * float(m00)) * dt) / dx)
        amps, sr, sc = None, 0, 0
        if source is not None:
            kind, sr, sc, fc, *extent = source
            sr = rows // 2 if sr is None else sr
            sc = cols // 2 if sc is None else sc
            if extent:
                eng.set_source_extent(*extent[0])
            f = {"ricker": ricker_amplitude, "sinusoidal": sinusoidal_amplitude}[kind]
            amps = np.array([f(i * dt, fc) for i in range(nsteps)], dtype=np.float64)
        every = max(1, nsteps // nframes) if on_frame is not None else nsteps
        done = 0
        while done < nsteps:
            if on_frame is None:
                n = nsteps - done
            else:                       # stop right after a step i with i % every == 0
                nxt = ((done + every - 1) // every) * every
                n = min(nsteps - done, nxt - done + 1)
            eng.run(n, sr, sc, None if amps is None else amps[done:done + n])
            done += n
            if on_frame is not None and (done - 1) % every == 0:
                Ez, _, _ = eng.download()
                on_frame(done - 1, Ez)
        return eng.download()
