#!/usr/bin/env python3
"""Does the point source cost time?  us per 16-step pass over a 400-step run with / without amplitudes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
for g in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4096").split(",")]:
    with fd.Engine(g, g, dtype=np.float32) as e:
        e.set_materials(); e.prepare(400); e.run(64).sync()
        amps = np.array([fd.ricker_amplitude(i * 5e-14, 30e9) for i in range(400)])
        z = np.zeros(400)
        for name, args in (("no source", (400,)), ("source at centre", (400, g // 2, g // 2, amps)),
                           ("zero amps at centre", (400, g // 2, g // 2, z)),
                           ("source at (c,100)", (400, g // 2, 100, amps)), ("source at (100,c)", (400, 100, g // 2, amps)),
                           ("source at (c+300,c+300)", (400, g // 2 + 300, g // 2 + 300, amps)),
                           ("source at (100,100)", (400, 100, 100, amps)), ("no source", (400,))):
            t = []
            for rep in range(3):
                e.timer_start(); e.run(*args); t.append(e.timer_stop())
            print(g, f"{name:22s} {min(t) * 1e3 / 25:8.1f} us per pass", "shape", e.last_shape, flush=True)
        ms = np.sort(e.time_launches(24, 16))
        print(g, f"{'time_launches':22s} {np.median(ms) * 1e3:8.1f} us per pass", flush=True)
