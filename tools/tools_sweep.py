#!/usr/bin/env python3
"""A/B sweep helper (GPU box): runs bench.py under env-selected library builds / knobs.
usage: tools_sweep.py "<lib>:<max_nt>:<band_rows>:<grid>[:materials]" ...   (lib '' = default)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:]:
    parts = spec.split(":")
    lib, nt, br, grid = parts[:4]
    mat = parts[4] if len(parts) > 4 else "uniform"
    env = dict(os.environ)
    if lib:
        env["FDTD2D_LIB"] = os.path.join(root, "fdtd-2d_amd", lib)
    env["FDTD2D_MAX_NT"] = nt
    env["FDTD2D_BAND_ROWS"] = br
    steps = "192" if int(grid) <= 8192 else "96"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--grid", grid, "--steps", steps,
                          "--warmup", "16", "--no-cpu-baseline", "--materials", mat],
                         env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{spec:40s} {d['value']:12.1f} Mcell-steps/s  frac {d['roofline']['frac']:.3f}  {d['ms_per_step']:.4f} ms/step", flush=True)
    except Exception:
        print(spec, "FAILED", out.stdout[-300:], out.stderr[-500:], flush=True)
