"""us per full-length pass for ROWS x COLS grids of equal cell count: does the row pitch matter (TLB reach)?
   python tools/ab_aspect.py R1xC1 R2xC2 ..."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
import bench
for spec in sys.argv[1:]:
    r, c = (int(v) for v in spec.split("x"))
    eng = bench.make_engine(fd, r, c, "uniform", 0, "mur")
    cyc = eng.cycle_steps
    eng.prepare(cyc * 4)
    eng.run(cyc * 3).sync()
    t = np.sort(eng.time_launches(32, cyc))
    print(f"{r}x{c}: us {t[2:-2].mean()*1e3:.1f} min {t[0]*1e3:.1f} ns/Mcell-16steps {t[2:-2].mean()*1e6/(r*c/1e6):.2f} copy {eng.measure_copy(4):.0f} GB/s shape {eng.last_shape}", flush=True)
    eng.close()
