"""us per full-length pass: the tuner's choice with and without filler-band candidates / for given shapes.
   python tools/ab_shape.py GRID [materials]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
import bench
grid = int(sys.argv[1]); mat = sys.argv[2] if len(sys.argv) > 2 else "uniform"
shapes = [tuple(int(v) for v in s.split(":")) for s in sys.argv[3:]]
eng = bench.make_engine(fd, grid, grid, mat, 0, "mur")
cyc = eng.cycle_steps
sr = sc = grid // 2
amps = bench.amplitudes(fd, 0, 64)
eng.prepare(cyc * 4, sr, sc)
eng.run(64, sr, sc, amps).sync()
def t(tag):
    eng.run(cyc, sr, sc, amps).sync()
    ms = np.sort([eng.timer_start() or eng.run(cyc, sr, sc, amps) and eng.timer_stop() for _ in range(24)])
    print(f"{grid} {mat} {tag}: us {ms[2:-2].mean()*1e3:.2f} min {ms[0]*1e3:.2f} shape {eng.last_shape}", flush=True)
for r in range(3):
    eng.set_shape((0,)); t("tuner")
    for s in shapes:
        eng.set_shape(s); t("given")
