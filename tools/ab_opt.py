"""A/B of an Engine.set_option knob in ONE process, alternating: us per full-length pass.
   python tools/ab_opt.py GRID name=a,b [materials] [rounds]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
import bench

grid = int(sys.argv[1]); name, vals = sys.argv[2].split("="); vals = [int(v) for v in vals.split(",")]
mat = sys.argv[3] if len(sys.argv) > 3 else "uniform"
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
eng = bench.make_engine(fd, grid, grid, mat, 0, "mur")
cyc = eng.cycle_steps
eng.run(64, grid // 2, grid // 2, bench.amplitudes(fd, 0, 64)).sync()
for v in vals:
    eng.set_option(**{name: v}); eng.prepare(cyc * 4); eng.run(cyc * 2).sync()
    print(name, v, "shape", eng.last_shape, flush=True)
for r in range(rounds):
    for v in vals:
        eng.set_option(**{name: v})
        eng.run(cyc).sync()
        t = np.sort(eng.time_launches(32, cyc))
        print(f"{grid} {mat} {name}={v} us {t[2:-2].mean()*1e3:.2f} min {t[0]*1e3:.2f} shape {eng.last_shape}", flush=True)
