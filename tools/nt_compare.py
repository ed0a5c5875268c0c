#!/usr/bin/env python3
"""us per step of full-length passes of 16 and 20 steps (steady state, tuned shapes).
    python tools/nt_compare.py 8192,16384"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
import bench
for g in [int(x) for x in sys.argv[1].split(",")]:
    eng = bench.make_engine(fd, g, g, "uniform", 0, "mur")
    for n in (16, 20):
        eng.prepare(n * 3); eng.run(n * 3).sync()
        ms = np.sort(eng.time_launches(24, n))
        print(f"{g}^2 run({n}): {np.median(ms) * 1e3:8.1f} us = {np.median(ms) * 1e3 / n:6.2f} us/step, shape {(eng.info(19), eng.info(20))}", flush=True)
    del eng
