"""`python -m fdtd2d_amd.fdtd` -- the experiment of the reference's python-src/fdtd.py:13-40 on
the GPU: rows = cols = 200, dt = 5e-14, dx = 1e-4, 1000 steps, ricker source (30 GHz) at the
centre, a snapshot every nsteps // nframes steps written as frames/frame_%04d.png with the
reference's colour scale (+-1e-3).  Differences, on purpose: the structure image is optional
(the reference's assets/example_structure.png is not in its repository; vacuum without it),
an existing frames/ directory is not deleted (the reference's main.py:7-9 does that on import),
and no video is encoded (main.py:126-150 shells out to ffmpeg)."""
from __future__ import annotations

import argparse
import os

import numpy as np

from . import api


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--rows", type=int, default=200)          # fdtd.py:14
    ap.add_argument("--cols", type=int, default=200)          # fdtd.py:15
    ap.add_argument("--dt", type=float, default=5e-14)        # fdtd.py:16
    ap.add_argument("--dx", type=float, default=1e-4)         # fdtd.py:17
    ap.add_argument("--nsteps", type=int, default=1000)       # fdtd.py:18
    ap.add_argument("--nframes", type=int, default=200)       # fdtd.py:19
    ap.add_argument("--structure", default=None, help="grayscale image -> eps (material_init)")
    ap.add_argument("--frames", default="frames", help="output directory ('' = no snapshots)")
    ap.add_argument("--dtype", choices=["float32", "float64"], default="float64")
    ap.add_argument("--boundary", choices=["mur", "pml"], default="mur")
    a = ap.parse_args(argv)

    eps, mu = api.material_init(a.structure, a.rows, a.cols)                     # fdtd.py:22
    print(api.courant_number(eps, mu, a.dt, a.dx))                               # fdtd.py:27
    every = max(1, a.nsteps // a.nframes)
    if a.frames:
        os.makedirs(a.frames, exist_ok=True)

    def on_frame(i, Ez):                                                         # fdtd.py:36-38
        api.capture_snapshot(Ez, eps, os.path.join(a.frames, f"frame_{i // every:04d}.png"), 1e-3, -1e-3)

    Ez, Hx, Hy = api.run_fdtd(a.rows, a.cols, a.dt, a.dx, a.nsteps, eps=eps, mu=mu,
                              source=("ricker", a.rows // 2, a.cols // 2, 30e9),
                              boundary=a.boundary, dtype=np.dtype(a.dtype),
                              on_frame=on_frame if a.frames else None, nframes=a.nframes)
    print(f"done: max|Ez| = {np.abs(Ez).max():.6g}, frames in {a.frames or '(none)'}")
    return Ez, Hx, Hy


if __name__ == "__main__":
    main()
