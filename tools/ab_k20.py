"""ms per run(20) (one 20-step pass) at GRID: python tools/ab_k20.py GRID"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
import bench
grid = int(sys.argv[1])
eng = bench.make_engine(fd, grid, grid, "uniform", 0, "mur")
sr = sc = grid // 2
amps = bench.amplitudes(fd, 0, 64)
eng.prepare(20, sr, sc)
eng.run(40, sr, sc, amps).sync()
ms = []
for _ in range(16):
    eng.timer_start(); eng.run(20, sr, sc, amps); ms.append(eng.timer_stop())
ms = np.sort(ms)
print(f"{grid} run(20): median {np.median(ms):.4f} ms min {ms[0]:.4f} shape {eng.last_shape} nt {eng.last_pass_steps} narrow={os.environ.get('FDTD2D_DEBUG_NARROW20')}", flush=True)
