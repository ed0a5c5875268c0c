#!/usr/bin/env python3
"""How much slower is ONE pass launched on an idle GPU than the same pass in a stream of them?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
g, n = int(sys.argv[1]), int(sys.argv[2])
sr = sc = g // 2
amps = np.array([fd.ricker_amplitude(i * 5e-14, 30e9) for i in range(n)])
with fd.Engine(g, g, dtype=np.float32) as e:
    e.set_materials(); e.prepare(n, sr, sc); e.run(n, sr, sc, amps).sync()
    for idle_ms in (0, 0, 1, 10, 100, 0):
        e.run(5, sr, sc, amps).sync()
        time.sleep(idle_ms / 1e3)
        t0 = time.perf_counter(); e.timer_start(); e.run(n, sr, sc, amps); ev = e.timer_stop(); e.sync()
        print(g, n, f"idle {idle_ms:4d} ms before: event {ev*1e3:8.1f} us  wall {(time.perf_counter()-t0)*1e6:8.1f} us", flush=True)
    ms = np.sort(e.time_launches(24, n))
    print(g, n, f"stream of launches (no source): median {np.median(ms)*1e3:.1f} us", flush=True)
    t = []
    for rep in range(5):
        e.timer_start(); e.run(n, sr, sc, amps); t.append(e.timer_stop())
    print(g, n, "back-to-back with source:", [round(x * 1e3, 1) for x in t], flush=True)
