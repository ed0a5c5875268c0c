#!/usr/bin/env python3
"""Interleaved A/B of band heights in one process: min / median us per 8-step pass."""
import os, sys, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
LS = int(os.environ.get("AB_LEVEL_SPLIT", "0"))
plan = {2048: (0, 16, 32, 48, 64), 4096: (0, 32, 48, 64, 96, 128), 8192: (0, 64, 96, 128, 192, 256),
        16384: (0, 128, 192, 256, 384, 512)} if LS else {2048: (0, 8, 16, 24, 32), 4096: (0, 20, 24, 28, 32, 36, 40, 48),
        8192: (0, 48, 64, 96, 128), 16384: (0, 64, 96, 128, 192)}
for g, bands in plan.items():
    res = {b: [] for b in bands}
    with fd.Engine(g, g, dtype=np.float32) as e:
        e.set_materials().set_option(level_split=LS); e.run(16); e.sync()
        n = 160 if g <= 8192 else 64
        for rnd in range(5):
            for b in bands:
                e.set_option(band_rows=b)
                e.run(8); e.sync()
                e.timer_start(); e.run(n); ms = e.timer_stop()
                res[b].append(ms / (n / 8) * 1000)
    print(g, " | ".join(f"{b}: {statistics.median(v):7.1f}" for b, v in res.items()), flush=True)
