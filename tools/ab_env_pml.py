"""A/B of an environment switch read at launch time (experiment builds) on one rank's PML slab, alternating in ONE process:
us per 16-step pair.   FDTD2D_LIB=build/x/libfdtd2d.so python tools/ab_env_pml.py ROWS COLS VAR [rounds]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
import bench
rows, cols, var = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
eng = bench.make_engine(fd, rows, cols, "uniform", 0, "pml")
cyc = eng.cycle_steps
os.environ[var] = "0"
eng.prepare(cyc * 4); eng.run(cyc * 8).sync()
for r in range(rounds):
    for v in ("0", "1"):
        os.environ[var] = v
        eng.run(cyc).sync()
        t = np.sort(eng.time_launches(24, cyc))
        print(f"{rows}x{cols} pml {var}={v} us {t[2:-2].mean()*1e3:.2f} min {t[0]*1e3:.2f} shape {eng.last_shape}", flush=True)
