"""us per full-length pass for forced (side_waves, xcd_map) combinations, tuner choosing the band heights; one process.
   python tools/ab_side.py GRID [materials] [rounds]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
import bench
grid = int(sys.argv[1]); mat = sys.argv[2] if len(sys.argv) > 2 else "uniform"; rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
eng = bench.make_engine(fd, grid, grid, mat, 0, "mur")
cyc = eng.cycle_steps
eng.run(64, grid // 2, grid // 2, bench.amplitudes(fd, 0, 64)).sync()
if os.environ.get("AB_RANDOM_FIELDS"):      # data in every cell: the chip holds a lower clock (profiles/r03_clock_vs_launch.txt)
    eng.upload(bench.hash_rows(0, grid, grid, 1, 1.0), bench.hash_rows(0, grid, grid, 2, 1e-3)[:, :grid - 1],
               bench.hash_rows(0, grid - 1, grid, 3, 1e-3))
    eng.run(32).sync()
    print("pseudo-random fields in every cell", flush=True)
combos = [(1, 0), (1, 1), (2, 0), (2, 1)] + ([(4, 0), (4, 1)] if mat == "uniform" else [])
for r in range(rounds):
    for sd, xc in combos:
        eng.set_option(side_waves=sd, xcd_map=xc)
        eng.run(cyc).sync()
        t = np.sort(eng.time_launches(32, cyc))
        print(f"{grid} {mat} side={sd} xcd={xc} us {t[2:-2].mean()*1e3:.2f} min {t[0]*1e3:.2f} shape {eng.last_shape}", flush=True)
eng.set_option(side_waves=0, xcd_map=-1)
eng.run(cyc).sync()
t = np.sort(eng.time_launches(32, cyc))
print(f"{grid} {mat} tuner's own choice: us {t[2:-2].mean()*1e3:.2f} min {t[0]*1e3:.2f} shape {eng.last_shape}", flush=True)
