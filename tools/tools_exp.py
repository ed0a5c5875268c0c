import os, sys, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
# us per 8-step pass (median of 7); experiment builds may give wrong fields by construction
for g in (2048, 4096, 8192):
    with fd.Engine(g, g, dtype=np.float32) as e:
        e.set_materials(); e.run(16); e.sync()
        v = np.sort(e.time_launches(40, 8))
        print(os.environ.get("FDTD2D_LIB", "default").split("/")[-1], g, f"{np.mean(v[4:-4]) * 1000:8.1f} us/pass", flush=True)
