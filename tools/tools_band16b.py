#!/usr/bin/env python3
"""Band-height sweep for 16-step passes on the per-GPU slab shapes of the multi-GPU bench."""
import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd

for (r, c) in ((4096, 8192), (4096, 16384), (4096, 32768), (2048, 8192), (6144, 6144)):
    with fd.Engine(r, c, dtype=np.float32) as e:
        e.set_materials(); e.set_option(max_pass_steps=16); e.run(32); e.sync()
        n = 320 if r * c <= 8192 * 8192 else 96
        out = []
        for nb in (6, 8, 12, 16, 20, 24, 32, 44, 64):
            br = math.ceil((r - 42) / nb)
            if br < 48: continue
            e.set_option(band_rows=br)
            e.run(16); e.sync()
            best = 1e9
            for rep in range(3):
                e.timer_start(); e.run(n); ms = e.timer_stop()
                best = min(best, ms / (n / 8) * 1000)
            out.append((nb, br, best))
        e.set_option(band_rows=0); e.run(16); e.sync()
        e.timer_start(); e.run(n); ms = e.timer_stop()
        print(f"{r}x{c}", "auto %.1f |" % (ms / (n / 8) * 1000),
              "  ".join(f"nb{nb}/br{br}: {t:.1f}" for nb, br, t in out), flush=True)
