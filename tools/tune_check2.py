#!/usr/bin/env python3
"""Which part of the tuner's protocol makes a shape look faster than it runs?  Real committed launches of a forced
shape, timed (a) one by one between HIP events, (b) six back to back between two events (the tuner's way)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
import bench
g, mat = int(sys.argv[1]), sys.argv[2]
eng = bench.make_engine(fd, g, g, mat, 0, "mur", autotune=False)
cyc = eng.cycle_steps
shapes = [tuple(int(x) for x in s.split("x")) for s in sys.argv[3:]]
amps = bench.amplitudes(fd, 0, cyc * 6) * 0
for rep in range(2):
    for sh in shapes:
        eng.set_option(long_shape=sh)
        eng.run(cyc * 2).sync()
        one = np.sort(eng.time_launches(24, cyc))
        six = []
        for k in range(4):
            eng.timer_start(); eng.run(cyc * 6); six.append(eng.timer_stop() / 6)
        six_src = []
        for k in range(4):
            eng.timer_start(); eng.run(cyc * 6, g // 2, g // 2, amps); six_src.append(eng.timer_stop() / 6)
        shape_src = eng.last_shape
        print(f"shape {sh}: one by one median {np.median(one) * 1e3:.1f} us (min {one[0] * 1e3:.1f}); six back to back "
              f"{min(six) * 1e3:.1f}; six with a zero source {min(six_src) * 1e3:.1f} (shape used {shape_src}, launches {eng.info(16)})", flush=True)
