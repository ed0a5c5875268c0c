#!/usr/bin/env python3
"""Launch-shape landscape of the full-length pass: us per launch over (band rows, waves per strip,
band rows of the first/last strip), shapes forced through FDTD2D_OPT_LONG_SHAPE.
    python tools/shape_sweep.py grid[,cols] materials "br:nw:er br:nw:er ..." """
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
import bench
g = [int(x) for x in sys.argv[1].split(",")]
rows, cols = g[0], g[-1]
mat = sys.argv[2]
shapes = [tuple(int(v) for v in s.split(":")) for s in sys.argv[3].split()]
eng = bench.make_engine(fd, rows, cols, mat, 0, "mur", autotune=False)
cyc = eng.cycle_steps
eng.run(cyc * 2).sync()
best = None
for rep in range(2):
    for sh in shapes:
        eng.set_option(long_shape=sh)
        eng.run(cyc * 2).sync()
        ms = np.sort(eng.time_launches(12, cyc))
        us = float(np.median(ms) * 1e3)
        if rep == 1:
            print(f"{rows}x{cols} {mat} shape {sh}: {us:8.1f} us  (min {ms[0]*1e3:.1f})", flush=True)
            if best is None or us < best[0]:
                best = (us, sh)
print("best", best)
