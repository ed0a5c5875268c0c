#!/usr/bin/env python3
"""Band-height sweep for 16-step passes (k_bulk_split<16>): us per 8 steps against the number
of workgroups = strips x bands, to calibrate the band heuristic (quantisation over 256 CUs)."""
import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd

def strips(cols, nt=16):
    return math.ceil((cols - 2 * 0) / (256 - 2 * nt))
for g, nbs in ((4096, (10, 13, 16, 20, 24, 26, 27, 30, 33, 36, 40, 41, 46, 50, 53, 54, 60, 64)),
               (8192, (6, 7, 8, 10, 13, 14, 16, 20, 21, 24, 27, 28, 32, 40, 48, 64)),
               (16384, (8, 10, 12, 13, 14, 16, 20, 24, 28, 32, 41, 48, 64))):
    with fd.Engine(g, g, dtype=np.float32) as e:
        e.set_materials(); e.set_option(max_pass_steps=16); e.run(32); e.sync()
        n = 320 if g <= 8192 else 96
        out = []
        for nb in nbs:
            br = math.ceil((g - 2 * 21) / nb)
            e.set_option(band_rows=br)
            e.run(16); e.sync()
            best = 1e9
            for rep in range(3):
                e.timer_start(); e.run(n); ms = e.timer_stop()
                best = min(best, ms / (n / 8) * 1000)
            out.append((nb, br, best))
        e.set_option(band_rows=0); e.run(16); e.sync()
        e.timer_start(); e.run(n); ms = e.timer_stop()
        print(g, "auto %.1f |" % (ms / (n / 8) * 1000),
              "  ".join(f"nb{nb}/br{br}/wg{nb*strips(g)}: {t:.1f}" for nb, br, t in out), flush=True)
