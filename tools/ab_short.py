#!/usr/bin/env python3
"""A/B of library builds on short passes: us per run(n), n = 1, 2, 4 (k_bulk) and 8 (k_bulk_split<8>).  python tools/ab_short.py libA.so,libB.so 4096"""
import os, subprocess, sys
libs = sys.argv[1].split(",")
g = sys.argv[2] if len(sys.argv) > 2 else "4096"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
import fdtd2d_amd as fd
g = int(sys.argv[1])
out = []
with fd.Engine(g, g, dtype=np.float32) as e:
    e.set_materials()
    for n in (1, 2, 4, 8):
        e.prepare(n); e.run(n); e.sync()
        ms = np.sort(e.time_launches(24, n))
        out.append(f"run({n}) {np.median(ms) * 1e3:.1f}")
print("RES", "  ".join(out))
''' % ROOT
for r in range(2):
    for lib in libs:
        env = dict(os.environ, FDTD2D_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, "-c", CHILD, g], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RES")]
        print(g, lib, line[-1] if line else ("FAILED " + p.stderr[-300:]), flush=True)
