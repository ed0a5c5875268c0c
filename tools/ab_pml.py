#!/usr/bin/env python3
"""A/B of library builds on the PML slab (one rank of BASELINE configs[4]): us per 16-step pass pair.
    python tools/ab_pml.py libA.so,libB.so [rows cols]"""
import os, subprocess, sys
libs = sys.argv[1].split(",")
rows, cols = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4096, 32768)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
import fdtd2d_amd as fd
import bench
eng = bench.make_engine(fd, int(sys.argv[1]), int(sys.argv[2]), "uniform", 0, "pml")
cyc = eng.cycle_steps
eng.prepare(cyc * 4); eng.run(cyc * 4).sync()
ms = np.sort(eng.time_launches(24, cyc))
print(json.dumps({"us": float(np.median(ms) * 1e3), "min": float(ms[0] * 1e3), "shape": list(eng.last_shape), "cyc": cyc}))
''' % ROOT
for r in range(2):
    for lib in libs:
        env = dict(os.environ, FDTD2D_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, "-c", CHILD, str(rows), str(cols)], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(rows, cols, lib, line[-1] if line else ("FAILED " + p.stderr[-300:]), flush=True)
