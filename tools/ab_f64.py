#!/usr/bin/env python3
"""A/B of library builds, float64 (the reference's dtype): us per 8-step pass.  python tools/ab_f64.py libA.so,libB.so 2048,4096"""
import os, subprocess, sys
libs = sys.argv[1].split(",")
grids = [int(g) for g in sys.argv[2].split(",")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
import fdtd2d_amd as fd
from fdtd2d_amd import _abi
if os.environ.get('AB_OLD_ABI'): _abi.SIGNATURES.pop('fdtd2d_rccl_selftest', None)
g = int(sys.argv[1])
eng = fd.Engine(g, g, 5e-14, 1e-4, dtype=np.float64)
eng.set_materials()
cyc = eng.cycle_steps
eng.prepare(cyc * 4); eng.run(cyc * 4).sync()
ms = np.sort(eng.time_launches(24, cyc))
print(json.dumps({"us": float(np.median(ms) * 1e3), "min": float(ms[0] * 1e3), "shape": list(eng.last_shape), "cyc": cyc}))
''' % ROOT
for g in grids:
    for r in range(2):
        for lib in libs:
            env = dict(os.environ, FDTD2D_LIB=os.path.abspath(lib), AB_OLD_ABI='1' if 'old' in lib else '')
            p = subprocess.run([sys.executable, "-c", CHILD, str(g)], env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            print(g, "float64", lib, line[-1] if line else ("FAILED " + p.stderr[-300:]), flush=True)
