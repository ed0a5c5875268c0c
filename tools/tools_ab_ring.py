#!/usr/bin/env python3
"""8-step (k_bulk) against 16-step (k_bulk_split<16,8> with coefficient rows) passes over array
materials: BASELINE configs[2] (8192^2 ring resonator, eps array) and eps+mu arrays at 4096^2."""
import os, sys, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
from oracle import fdtd_numpy as onp

for (g, kind) in ((4096, "ring"), (8192, "ring"), (8192, "eps+mu"), (12288, "ring")):
    eps = onp.ring_resonator_eps(g, g).astype(np.float32)
    mu = np.full((g, g), onp.MU0, np.float32)
    if kind == "eps+mu":
        mu[g // 4: g // 2] *= 2
    res = {}
    with fd.Engine(g, g, dtype=np.float32) as e:
        e.set_materials(eps, mu); e.run(32); e.sync()
        for rnd in range(5):
            for name, opt in (("8-step", dict(max_pass_steps=8, band_rows=0, split_waves=0)),
                              ("16-step", dict(max_pass_steps=16, band_rows=0, split_waves=0)),
                              ("16/nw8/b96", dict(max_pass_steps=16, band_rows=96, split_waves=8)),
                              ("16/nw8/b160", dict(max_pass_steps=16, band_rows=160, split_waves=8)),
                              ("16/nw4/b0", dict(max_pass_steps=16, band_rows=0, split_waves=4)),
                              ("16/nw4/b96", dict(max_pass_steps=16, band_rows=96, split_waves=4)),
                              ("16/nw4/b160", dict(max_pass_steps=16, band_rows=160, split_waves=4)),
                              ("16/nw4/b256", dict(max_pass_steps=16, band_rows=256, split_waves=4))):
                e.set_option(**opt)
                e.run(16); e.sync()
                e.timer_start(); e.run(160); ms = e.timer_stop()
                res.setdefault(name, []).append(ms / 20 * 1000)
    print(g, kind, "  ".join(f"{k}: min {min(v):7.1f} med {statistics.median(v):7.1f}" for k, v in res.items()),
          f" Tcs/s(16) {g*g*8/min(res['16-step'])/1e6:.3f}", flush=True)
