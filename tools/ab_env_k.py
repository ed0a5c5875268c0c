"""A/B of an environment switch read at launch time, alternating in ONE process: ms per run(K).
   FDTD2D_LIB=... python tools/ab_env_k.py GRID VAR K [rounds]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
import bench
grid = int(sys.argv[1]); var = sys.argv[2]; K = int(sys.argv[3]); rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
eng = bench.make_engine(fd, grid, grid, "uniform", 0, "mur")
sr = sc = grid // 2
amps = bench.amplitudes(fd, 0, 64)
os.environ[var] = "0"
eng.prepare(K, sr, sc)
eng.run(64, sr, sc, amps).sync()
for r in range(rounds):
    for v in ("0", "1"):
        os.environ[var] = v
        ms = []
        for _ in range(12):
            eng.timer_start(); eng.run(K, sr, sc, amps); ms.append(eng.timer_stop())
        ms = np.sort(ms)
        print(f"{grid} run({K}) {var}={v}: median {np.median(ms):.4f} ms min {ms[0]:.4f} shape {eng.last_shape}", flush=True)
