import os, sys, time, subprocess
# wall time per pass with experiment libs (results are wrong by construction; bench asserts finite field)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
for g in (4096, 16384):
    with fd.Engine(g, g, dtype=np.float32) as e:
        e.set_materials(); e.run(16); e.sync()
        n = 160 if g == 4096 else 48
        e.timer_start(); e.run(n); ms = e.timer_stop()
        print(os.environ.get("FDTD2D_LIB", "default").split("/")[-1], g, f"{ms / (n / 8) * 1000:8.1f} us/pass")
