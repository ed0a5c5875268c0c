#!/usr/bin/env python3
"""Does the tuner's view (uncommitted trial launches, same buffers every time) rank launch shapes like real,
committed, ping-ponging launches do?  FDTD2D_LIB=build/tunelog/libfdtd2d.so python tools/tune_check.py 8192 ring"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
import bench
g, mat = int(sys.argv[1]), sys.argv[2]
eng = bench.make_engine(fd, g, g, mat, 0, "mur")
cyc = eng.cycle_steps
eng.prepare(cyc * 4, g // 2, g // 2)
eng.run(cyc * 4, g // 2, g // 2, bench.amplitudes(fd, 0, cyc * 4)).sync()
print("tuner picked", eng.last_shape, flush=True)
shapes = [tuple(int(x) for x in s.split("x")) for s in sys.argv[3:]] or [eng.last_shape]
for rep in range(2):
    for sh in shapes:
        eng.set_option(long_shape=sh)
        eng.run(cyc * 2).sync()
        ms = np.sort(eng.time_launches(24, cyc))
        print(f"real launches, shape {sh}: median {np.median(ms) * 1e3:.1f} us, min {ms[0] * 1e3:.1f}", flush=True)
