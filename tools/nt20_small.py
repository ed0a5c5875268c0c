#!/usr/bin/env python3
"""run(20) as one 20-step pass (max_pass_steps = 20 lifts the size rule) vs the default plan (16 + 4 below 64 Mi cells)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
import bench
for g in [int(x) for x in sys.argv[1].split(",")]:
    for forced in (False, True):
        eng = bench.make_engine(fd, g, g, "uniform", 0, "mur")
        if forced:
            eng.set_option(max_pass_steps=20)
        amps = bench.amplitudes(fd, 0, 20)
        eng.prepare(20, g // 2, g // 2); eng.run(20, g // 2, g // 2, amps).sync()
        t = []
        for k in range(12):
            l0 = eng.info(16)
            eng.timer_start(); eng.run(20, g // 2, g // 2, amps); t.append(eng.timer_stop())
            n = eng.info(16) - l0
        print(f"{g}^2 run(20) {'one 20-step pass (forced)' if forced else 'default plan'}: median {np.median(t) * 1e3:.1f} us, min {min(t) * 1e3:.1f}, {n} pass launches, shape {eng.last_shape}", flush=True)
        eng.close()
