#!/usr/bin/env python3
"""Full-size parity evidence (GPU box): BASELINE config 2 at its full 2000 steps and config 3's
8192^2 ring-resonator grid for 200 steps, device (temporally blocked passes) vs the OpenMP C
oracle on the host, np.array_equal on all three fields.  Too slow for the pytest suite (the
oracle needs minutes); its output is kept in profiles/r01_full_config_parity.txt."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd
from oracle import c_oracle, fdtd_numpy as onp

DT, DX, FC = 5e-14, 1e-4, 30e9
for name, n, steps, eps_fn in (("config 2: 4096x4096 uniform", 4096, 2000, None),
                               ("config 3 grid: 8192x8192 ring eps", 8192, 200, onp.ring_resonator_eps)):
    eps = (np.full((n, n), onp.EPS0) if eps_fn is None else eps_fn(n, n)).astype(np.float32)
    mu = np.full((n, n), onp.MU0, np.float32)
    sr, sc = (n // 2, n // 2) if eps_fn is None else (int(0.2 * n), int(0.2 * n))
    amps = np.array([onp.ricker_amplitude(i * DT, FC) for i in range(steps)])
    t0 = time.time()
    with fd.Engine(n, n, DT, DX, dtype=np.float32) as eng:
        eng.set_materials(eps, mu)
        eng.run(steps, sr, sc, amps)
        got = eng.download()
        launches = eng.info(16)
        shape = (eng.cycle_steps, eng.info(19), eng.info(20))
    t1 = time.time()
    ref = onp.grid_zeros(n, n, np.float32)
    for d in range(0, steps, 100):                       # progress lines keep the run alive
        k = min(100, steps - d)
        c_oracle.run(*ref, eps, mu, DT, DX, k, sr, sc, amps=amps[d:d + k])
        print(f"  oracle at step {d + k} ({time.time() - t1:.0f}s)", flush=True)
    t2 = time.time()
    same = [bool(np.array_equal(a, b)) for a, b in zip(got, ref)]
    print(f"{name}, {steps} steps: device {t1 - t0:.1f}s ({launches} pass launches; {shape[0]}-step "
          f"passes, bands of {shape[1]} rows, {shape[2]} waves per strip), C oracle "
          f"({c_oracle.num_threads()} threads) {t2 - t1:.1f}s, max|Ez| {np.abs(ref[0]).max():.6g}, "
          f"Ez/Hx/Hy identical: {same}", flush=True)
    assert all(same)
