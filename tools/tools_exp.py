import os, sys, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
# us per 8-step pass, level-split forced on/off, a few band heights
for g, bands in ((2048, (0, 32, 64)), (4096, (0, 32, 48, 64, 96)), (8192, (0, 64, 128)), (16384, (0, 128, 256))):
    with fd.Engine(g, g, dtype=np.float32) as e:
        e.set_materials(); e.run(16); e.sync()
        out = []
        for ls in (0, 1):
            for b in bands:
                e.set_option(level_split=ls, band_rows=b); e.run(8); e.sync()
                v = np.sort(e.time_launches(24, 8))
                out.append(f"ls{ls}/b{b}: {np.mean(v[3:-3]) * 1000:7.1f}")
        print(os.environ.get("FDTD2D_LIB", "default").split("/")[-1], g, " | ".join(out), flush=True)
