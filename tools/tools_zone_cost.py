#!/usr/bin/env python3
"""How much of a 16-step pass is the zone tiles?  Times the bulk rows alone (a partial pass over
rows [21, R-21), dropped afterwards) against the whole pass; float32 uniform."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
for g in (2048, 4096, 8192):
    with fd.Engine(g, g, dtype=np.float32) as e:
        e.set_materials(); e.set_option(max_pass_steps=16); e.prepare(64); e.run(64); e.sync()
        full = np.median(e.time_launches(24, 16)) * 1000
        br, nw = e.info(19), e.info(20)
        e.set_option(band_rows=br, split_waves=nw)
        ts = []
        for rep in range(24):
            e.sync(); e.timer_start()
            e.pass_rows(16, 21, g - 21)
            ts.append(e.timer_stop() * 1000)
            try:
                e.pass_commit()
            except fd.Fdtd2dError:
                pass
        print(g, f"whole pass {full:.1f} us | strips only {np.median(ts):.1f} us | shape ({br}, {nw})", flush=True)
