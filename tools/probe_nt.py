#!/usr/bin/env python3
"""us per run(n) for forced band heights: python tools/probe_nt.py grid n "br br ..." """
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
g, n = int(sys.argv[1]), int(sys.argv[2])
with fd.Engine(g, g, dtype=np.float32) as e:
    e.set_materials()
    for br in [int(b) for b in sys.argv[3].split()]:
        e.set_option(band_rows=br, autotune=False)
        e.run(n); e.sync()
        ms = np.sort(e.time_launches(12, n))
        print(g, n, "band_rows", br, f"{np.median(ms)*1e3:.1f} us", "shape", e.last_shape, flush=True)
