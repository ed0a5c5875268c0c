#!/usr/bin/env python3
"""Interleaved A/B timing in ONE process (guide rule 24): for each grid, alternate option sets over
several rounds and report min / median us per 8-step pass (HIP events on the engine's stream)."""
import os, sys, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd

variants = {"k_bulk": dict(level_split=0), "k_bulk_split": dict(level_split=1)}
dtype = np.float64 if "f64" in sys.argv[1:] else np.float32
for g in (1024, 2048, 3072, 4096, 6144, 8192) + (() if dtype == np.float64 else (16384,)):
    res = {k: [] for k in variants}
    with fd.Engine(g, g, dtype=dtype) as e:
        if "arr" in sys.argv[1:]:
            from oracle import fdtd_numpy as onp
            e.set_materials(onp.ring_resonator_eps(g, g).astype(dtype), np.full((g, g), onp.MU0, dtype))
        else:
            e.set_materials()
        e.set_option(max_pass_steps=8); e.run(16); e.sync()
        n = 160 if g <= 8192 else 64
        for rnd in range(7):
            for name, opt in variants.items():
                e.set_option(**opt)
                e.run(8); e.sync()
                e.timer_start(); e.run(n); ms = e.timer_stop()
                res[name].append(ms / (n / 8) * 1000)
    print(g, "  ".join(f"{k}: min {min(v):8.1f} med {statistics.median(v):8.1f} us/pass" for k, v in res.items()), flush=True)
