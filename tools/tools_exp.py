import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
# us per 8 steps: 8-step passes (auto kernel) vs 16-step passes (level-split), a few band heights
for g, bands in ((4096, (0, 64, 128)), (8192, (0, 128, 256)), (16384, (0, 128, 256, 512))):
    with fd.Engine(g, g, dtype=np.float32) as e:
        e.set_materials(); e.run(16); e.sync()
        out = []
        for nt in (8, 16):
            for b in bands if nt == 16 else (0,):
                e.set_option(max_pass_steps=nt, band_rows=b); e.run(nt); e.sync()
                v = np.sort(e.time_launches(20, nt))
                out.append(f"nt{nt}/b{b}: {np.mean(v[3:-3]) * 1000 * 8 / nt:7.1f}")
        print(g, "us per 8 steps:", " | ".join(out), flush=True)
