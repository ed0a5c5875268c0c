#!/usr/bin/env python3
"""A/B of library builds: us per 16-step pass (median of 24 launches after tuning) for each
(library, grid), alternating, one subprocess per measurement.
    python tools/ab_lib.py libA.so,libB.so 4096,8192,16384 [materials] [rounds]"""
import os, subprocess, sys, json
libs = sys.argv[1].split(",")
grids = [int(g) for g in sys.argv[2].split(",")]
mat = sys.argv[3] if len(sys.argv) > 3 else "uniform"
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tools"))
import fdtd2d_amd as fd
from fdtd2d_amd import _abi
if os.environ.get('AB_OLD_ABI'): _abi.SIGNATURES.pop('fdtd2d_rccl_selftest', None)
import bench
g, mat = int(sys.argv[1]), sys.argv[2]
eng = bench.make_engine(fd, g, g, mat, 0, "mur")
cyc = eng.cycle_steps
eng.prepare(cyc * 4); eng.run(cyc * 4).sync()
ms = np.sort(eng.time_launches(24, cyc))
print(json.dumps({"us": float(np.median(ms) * 1e3), "min": float(ms[0] * 1e3), "shape": list(eng.last_shape), "cyc": cyc}))
''' % (ROOT, ROOT)
for g in grids:
    for r in range(rounds):
        for lib in libs:
            env = dict(os.environ, FDTD2D_LIB=os.path.abspath(lib), AB_OLD_ABI='1' if '/old/' in lib else '')
            p = subprocess.run([sys.executable, "-c", CHILD, str(g), mat], env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            print(g, mat, lib, line[-1] if line else ("FAILED " + p.stderr[-300:]), flush=True)
