"""A/B of an environment switch read at launch time (experiment builds), alternating in ONE process: us per full-length pass.
   FDTD2D_LIB=build/x/libfdtd2d.so python tools/ab_env.py GRID VAR [materials] [rounds]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
import bench
grid = int(sys.argv[1]); var = sys.argv[2]; mat = sys.argv[3] if len(sys.argv) > 3 else "uniform"; rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 6
eng = bench.make_engine(fd, grid, grid, mat, 0, "mur")
cyc = eng.cycle_steps
os.environ[var] = "0"
eng.prepare(cyc * 4); eng.run(cyc * 8).sync()
for r in range(rounds):
    for v in ("0", "1"):
        os.environ[var] = v
        eng.run(cyc).sync()
        t = np.sort(eng.time_launches(24, cyc))
        print(f"{grid} {mat} {var}={v} us {t[2:-2].mean()*1e3:.2f} min {t[0]*1e3:.2f} shape {eng.last_shape}", flush=True)
