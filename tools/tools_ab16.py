#!/usr/bin/env python3
"""Interleaved A/B in ONE process: 8-step passes (k_bulk / k_bulk_split<8>, automatic choice)
against 16-step passes (k_bulk_split<16>), float32 + uniform materials.  us per 8 steps."""
import os, sys, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd

variants = {"8-step": dict(max_pass_steps=8), "16-step": dict(max_pass_steps=16)}
shapes = [(g, g) for g in (256, 512, 1024, 2048, 3072, 4096, 6144, 8192, 16384)] + [(4096, 32768)]
for r, c in shapes:
    res = {k: [] for k in variants}
    with fd.Engine(r, c, dtype=np.float32) as e:
        e.set_materials(); e.run(32); e.sync()
        n = 320 if r * c <= 8192 * 8192 else 96
        for rnd in range(7):
            for name, opt in variants.items():
                e.set_option(**opt)
                e.run(16); e.sync()
                e.timer_start(); e.run(n); ms = e.timer_stop()
                res[name].append(ms / (n / 8) * 1000)
    print(f"{r}x{c}", "  ".join(f"{k}: min {min(v):8.1f} med {statistics.median(v):8.1f}" for k, v in res.items()),
          f"  Tcs/s(16) {r*c*8/min(res['16-step'])/1e6:.3f}", flush=True)
