import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
# us per 8-step pass for (grid, band_rows) pairs; experiment builds give wrong fields by construction
for g, br in ((1024, 128), (1024, 32), (1024, 8), (4096, 0)):
    with fd.Engine(g, g, dtype=np.float32) as e:
        e.set_materials().set_option(band_rows=br); e.run(16); e.sync()
        n = 160
        e.timer_start(); e.run(n); ms = e.timer_stop()
        print(os.environ.get("FDTD2D_LIB", "default").split("/")[-1], g, br, f"{ms / (n / 8) * 1000:8.1f} us/pass", flush=True)
