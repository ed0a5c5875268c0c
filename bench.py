#!/usr/bin/env python3
"""Benchmark of the FDTD hot path on MI355X.  Prints ONE JSON line (rank 0).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--grid R] [--materials uniform|array]

A "step" is one full leapfrog step (H half-step, E half-step with Mur frame, point source)
over the whole grid.  The N=1 workload is BASELINE.json configs[1]: 4096x4096 fp32 TE-mode,
uniform eps (synthetic: vacuum, ricker source at the centre).  Fields are zero-initialised
in HBM before the timed region; only the per-step source amplitude crosses from the host.

Reported: metric value = Mcell-steps/s (whole job); roofline = algorithmic bytes per step
/ average step time measured with HIP events on the engine's stream, against the 8 TB/s
HBM3E peak; cpu_baseline = the NumPy oracle (a line-for-line structural stand-in for the
reference's NumPy code) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DT, DX, FC = 5e-14, 1e-4, 30e9
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(grid: int, budget_s: float = 12.0):
    """NumPy oracle (single core, like the reference's NumPy loop) on the same workload,
    bounded to ~budget_s; plus the multi-threaded C oracle for context."""
    from oracle import fdtd_numpy as onp
    from oracle import c_oracle
    g = min(grid, 4096)
    Ez, Hx, Hy = onp.grid_zeros(g, g, np.float32)
    eps, mu = onp.vacuum_materials(g, g, np.float32)
    onp.leapfrog(Ez, Hx, Hy, eps, mu, DT, DX, 1, g // 2, g // 2)      # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        onp.leapfrog(Ez, Hx, Hy, eps, mu, DT, DX, 1, g // 2, g // 2, step0=n + 1)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 64:
            break
    val = g * g * n / el / 1e6
    out = {"value": round(val, 2), "unit": "Mcell-steps/s", "cores": 1, "kind": "port",
           "sample": f"NumPy oracle, {g}x{g} fp32 vacuum, {n} steps in {el:.1f}s"}
    try:
        Ez, Hx, Hy = onp.grid_zeros(g, g, np.float32)
        c_oracle.run(Ez, Hx, Hy, eps, mu, DT, DX, 1, g // 2, g // 2)
        t0 = time.perf_counter()
        k = 8
        c_oracle.run(Ez, Hx, Hy, eps, mu, DT, DX, k, g // 2, g // 2)
        el = time.perf_counter() - t0
        out["c_port"] = {"value": round(g * g * k / el / 1e6, 2), "cores": c_oracle.num_threads(),
                         "sample": f"C oracle (OpenMP), {g}x{g} fp32, {k} steps in {el:.2f}s"}
    except Exception as exc:  # the C oracle is optional context
        out["c_port"] = {"error": str(exc)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--materials", choices=["uniform", "array"], default="uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import fdtd2d_amd as fd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        sys.exit("multi-GPU bench not wired yet in this revision")
    torch.cuda.set_device(local)

    n = args.grid
    amps_w = np.array([fd.ricker_amplitude(i * DT, FC) for i in range(args.warmup)])
    amps = np.array([fd.ricker_amplitude((args.warmup + i) * DT, FC) for i in range(args.steps)])
    eng = fd.Engine(n, n, DT, DX, dtype=np.float32, device=local)
    if args.materials == "uniform":
        eng.set_materials()
    else:
        eps, mu = fd.material_init(None, n, n)
        eng.set_materials(eps.astype(np.float32), mu.astype(np.float32), allow_uniform=False)
    eng.run(args.warmup, n // 2, n // 2, amps_w).sync()

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.timer_start()
    eng.run(args.steps, n // 2, n // 2, amps)
    ev_ms = eng.timer_stop()
    eng.sync()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0

    cells = n * n
    bpc = eng.bytes_per_cell_step
    value = cells * args.steps / wall / 1e6
    step_ms = ev_ms / args.steps
    achieved = cells * bpc / (step_ms * 1e-3) / 1e9
    Ez, _, _ = eng.download()
    assert np.isfinite(Ez).all() and np.abs(Ez).max() > 0
    res = {
        "metric": "Mcell-steps/s", "value": round(value, 1), "unit": "Mcell-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall * 1e3 / args.steps, 5), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{n}x{n} fp32 TE-mode, {args.materials} eps/mu, Mur-5 boundary, "
                               "ricker point source at centre (BASELINE configs[1])",
                   "grid": [n, n], "materials": args.materials, "kernel_path": "two-kernel step"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "bytes_per_cell_step": bpc, "launch": "one leapfrog step (H + E + frame + source kernels)",
                     "avg_launch_ms": round(step_ms, 5)},
    }
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(n)
    eng.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
