#!/usr/bin/env python3
"""Fixed launch-shape rules against the measured shape (autotune), interleaved in one process:
us per 8 steps, float32.  Prints the shape the tuner kept."""
import os, sys, statistics, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
from oracle import fdtd_numpy as onp

cases = [(4096, 4096, "uniform"), (6144, 6144, "uniform"), (8192, 8192, "uniform"), (8192, 8192, "ring"),
         (16384, 16384, "uniform"), (4096, 8192, "uniform"), (4096, 32768, "uniform"), (3072, 3072, "uniform"),
         (4096, 4096, "ring")]
for r, c, kind in cases:
    res = {"rules": [], "tuned": []}
    with fd.Engine(r, c, dtype=np.float32) as e:
        if kind == "ring":
            e.set_materials(onp.ring_resonator_eps(r, c).astype(np.float32), np.full((r, c), onp.MU0, np.float32))
        else:
            e.set_materials()
        e.set_option(autotune=0); e.run(32); e.sync()
        n = 320 if r * c <= 8192 * 8192 else 96
        t0 = time.perf_counter(); e.set_option(autotune=1); e.run(16); e.sync(); tune_s = time.perf_counter() - t0
        shape = (e.info(19), e.info(20))
        for rnd in range(5):
            for name, on in (("rules", 0), ("tuned", 1)):
                e.set_option(autotune=on)
                e.run(16); e.sync()
                e.timer_start(); e.run(n); ms = e.timer_stop()
                res[name].append(ms / (n / 8) * 1000)
        e.set_option(autotune=0); e.run(16); rule_shape = (e.info(19), e.info(20))
    print(f"{r}x{c} {kind}: rules {rule_shape} min {min(res['rules']):7.1f} med {statistics.median(res['rules']):7.1f} | "
          f"tuned {shape} min {min(res['tuned']):7.1f} med {statistics.median(res['tuned']):7.1f} | "
          f"tuning {tune_s*1e3:.0f} ms | {r*c*8/min(res['tuned'])/1e6:.3f} Tcs/s", flush=True)
