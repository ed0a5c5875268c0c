#!/usr/bin/env python3
"""k_bulk_split with 4 or 8 waves per strip: us per 8 steps over band heights (float32 uniform)."""
import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd

def t(e, n):
    e.run(32); e.sync()
    best = 1e9
    for rep in range(3):
        e.timer_start(); e.run(n); ms = e.timer_stop()
        best = min(best, ms / (n / 8) * 1000)
    return best

for g in (512, 1024, 2048, 3072, 4096):
    with fd.Engine(g, g, dtype=np.float32) as e:
        e.set_materials(); e.run(32); e.sync()
        n = 320
        for nt in (8, 16):
            e.set_option(max_pass_steps=nt, level_split=1)
            line = []
            for nw in (4, 8):
                e.set_option(split_waves=nw)
                for br in (0, 32, 48, 64, 96, 128, 160, 200, 256, 340):
                    if nt == 16 and 0 < br < 64: continue
                    if br > g // 8: continue
                    e.set_option(band_rows=br)
                    line.append(f"nw{nw}/b{br}: {t(e, n):.1f}")
            print(g, f"nt{nt}", "  ".join(line), flush=True)
