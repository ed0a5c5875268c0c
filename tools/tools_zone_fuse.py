#!/usr/bin/env python3
"""Level-split passes: zone tiles fused into the k_bulk_split launch (zone_split=0) against
k_zone on the side stream (zone_split=1); interleaved A/B, us per 8 steps, fixed launch rules."""
import os, sys, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd

for g in (512, 1024, 2048, 3072, 4096, 6144, 8192, 16384):
    with fd.Engine(g, g, dtype=np.float32) as e:
        e.set_materials(); e.set_option(autotune=0); e.run(32); e.sync()
        n = 320 if g <= 8192 else 96
        out = []
        for nt in (8, 16):
            res = {0: [], 1: []}
            e.set_option(max_pass_steps=nt, level_split=1)
            for rnd in range(5):
                for zs in (0, 1):
                    e.set_option(zone_split=zs)
                    e.run(16); e.sync()
                    e.timer_start(); e.run(n); ms = e.timer_stop()
                    res[zs].append(ms / (n / 8) * 1000)
            out.append(f"nt{nt}: fused {min(res[0]):7.1f} side {min(res[1]):7.1f}")
    print(g, " | ".join(out), flush=True)
