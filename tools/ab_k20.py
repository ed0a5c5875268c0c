"""ms per run(20) (one 20-step pass) at GRID, tuner's own shape and then the same shape with entry 7 (zone tiles fused into the
bulk launch) flipped, alternating: python tools/ab_k20.py GRID [uniform|ring|array]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
import bench
grid = int(sys.argv[1]); mat = sys.argv[2] if len(sys.argv) > 2 else "uniform"
eng = bench.make_engine(fd, grid, grid, mat, 0, "mur")
sr = sc = grid // 2
amps = bench.amplitudes(fd, 0, 64)
eng.prepare(20, sr, sc)
eng.run(40, sr, sc, amps).sync()
def t(tag):
    ms = []
    for _ in range(16):
        eng.timer_start(); eng.run(20, sr, sc, amps); ms.append(eng.timer_stop())
    ms = np.sort(ms)
    print(f"{grid} {mat} run(20) {tag}: median {np.median(ms):.4f} ms min {ms[0]:.4f} shape {eng.last_shape} nt {eng.last_pass_steps}", flush=True)
t("tuner")
own = list(eng.last_shape)
if own[3] == 1:
    other = own[:7] + [1 - own[7]]
    for _ in range(3):
        eng.set_shape(other, 20); t("flipped")
        eng.set_shape(own, 20); t("tuner's")
