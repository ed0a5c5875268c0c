#!/usr/bin/env python3
"""Cost of run(n) for every pass length: us per run(n) (HIP events, median of 24), float32 uniform.
n <= 24 is ONE pass (the shortest kernel that holds n levels); one sweep over the grid each.
    python tools/tools_short.py [grid,grid,...] [n,n,...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
grids = [int(g) for g in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4096, 8192]
ns = [int(n) for n in sys.argv[2].split(",")] if len(sys.argv) > 2 else [24, 20, 16, 15, 12, 9, 7, 5, 3, 8, 4, 2, 1]
for g in grids:
    with fd.Engine(g, g, dtype=np.float32) as e:
        e.set_materials()
        out = []
        for n in ns:
            try:
                e.prepare(n); e.run(n); e.sync()
                ms = np.sort(e.time_launches(24, n))
                out.append(f"run({n}): {np.median(ms)*1000:.1f} [{np.median(ms)*1000/n:.2f}/step]")
            except Exception as exc:
                out.append(f"run({n}): {type(exc).__name__}")
        print(g, "  ".join(out), flush=True)
