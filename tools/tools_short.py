#!/usr/bin/env python3
"""Cost of a short pass (the 16-step kernel stopping after n levels) against full passes and the
power-of-two decomposition, 4096^2 float32 uniform: us per run(n), HIP events, median of 24."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
for g in (4096, 8192):
    with fd.Engine(g, g, dtype=np.float32) as e:
        e.set_materials(); e.prepare(64); e.run(64); e.sync()
        out = []
        for n in (16, 15, 12, 9, 7, 5, 3, 8, 4, 2, 1):
            ms = np.sort(e.time_launches(24, n))
            out.append(f"run({n}): {np.median(ms)*1000:.1f}")
        print(g, "  ".join(out), flush=True)
