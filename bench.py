#!/usr/bin/env python3
"""Benchmark of the FDTD hot path on MI355X.  Prints ONE JSON line (rank 0).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--grid R [--cols C]]
                  [--materials uniform|array|ring] [--no-cpu-baseline]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full leapfrog step (H half-step, E half-step with the Mur frame, point
source) over the whole grid -- fdtd.py:31-34 of the reference.

Workloads (all synthetic: zero fields, ricker source at the grid centre, fp32):
  N = 1  BASELINE.json configs[1]: 4096 x 4096, uniform eps (vacuum), Mur-5 boundary.
  N > 1  row slabs of 4096 rows per GPU, columns 4096*N, i.e. the grids of BASELINE
         configs[3] (16384^2 on 4) and configs[4] (32768^2 on 8; Mur-5 frame); one process
         per GPU, halo exchange over torch.distributed (RCCL).  The same slab shape is also
         timed on one GPU alone ("single_gpu_same_slab") so that weak-scaling efficiency can
         be read from one line.

Timing: inputs resident in HBM; W untimed warm-up steps; barrier + device sync; K steps;
device sync + barrier; max over ranks.  value = cells * K / time.

roofline: the dominant kernel is the temporally blocked pass (k_bulk_split: one launch = 16
time steps over the whole slab for float32 + uniform materials; k_bulk: 8 steps otherwise).
achieved = algorithmic bytes per launch / average launch duration, where algorithmic bytes
= cells * steps per launch * (24 B + 4 B per non-uniform coefficient array)
and the duration is the trimmed mean of 48 back-to-back launches, each between its own pair
of HIP events on the engine's stream (rocprofv3's per-kernel average agrees: profiles/r01f_*.txt);
the per-launch share of the whole timed region, gaps included, is reported beside it.  Because a pass keeps 8 or 16 time
levels on chip, the algorithmic rate may exceed the HBM peak: frac > 1 means the kernel moves
fewer bytes than a one-step-per-pass scheme has to ("traffic" holds the measured bytes).
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
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
SLAB_ROWS = 4096


def cpu_baseline(grid: int, budget_s: float = 12.0):
    """The NumPy oracle -- expression for expression the structure of the reference's NumPy
    code, single core like it -- on the N=1 workload, bounded to ~budget_s; plus the
    OpenMP C oracle on all host cores for context.  Baseline, not target."""
    from oracle import c_oracle
    from oracle import fdtd_numpy as onp
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
    out = {"value": round(g * g * n / el / 1e6, 2), "unit": "Mcell-steps/s", "cores": 1,
           "kind": "port",
           "sample": f"NumPy oracle (oracle/fdtd_numpy.py), {g}x{g} fp32 vacuum, {n} steps in {el:.1f}s"}
    try:
        Ez, Hx, Hy = onp.grid_zeros(g, g, np.float32)
        c_oracle.run(Ez, Hx, Hy, eps, mu, DT, DX, 1, g // 2, g // 2)
        k, t0 = 8, time.perf_counter()
        c_oracle.run(Ez, Hx, Hy, eps, mu, DT, DX, k, g // 2, g // 2)
        el = time.perf_counter() - t0
        out["c_port"] = {"value": round(g * g * k / el / 1e6, 2), "cores": c_oracle.num_threads(),
                         "sample": f"C oracle (OpenMP), {g}x{g} fp32, {k} steps in {el:.2f}s"}
    except Exception as exc:  # context only
        out["c_port"] = {"error": str(exc)}
    return out


def amplitudes(fd, first, n):
    return np.array([fd.ricker_amplitude((first + i) * DT, FC) for i in range(n)])


def make_materials(fd, kind, rows, cols, r0=0, r1=None):
    """eps, mu for global rows [r0, r1) -- scalars for the uniform case."""
    r1 = rows if r1 is None else r1
    if kind == "uniform":
        return None, None
    if kind == "array":            # uniform values handed over as full arrays, detection off
        return (np.full((r1 - r0, cols), fd.EPS0, np.float32),
                np.full((r1 - r0, cols), fd.MU0, np.float32))
    if kind == "ring":             # BASELINE configs[2] geometry
        i = np.arange(r0, r1, dtype=np.float64)[:, None]
        j = np.arange(cols, dtype=np.float64)[None, :]
        core = (i >= np.floor(0.18 * rows)) & (i < np.floor(0.22 * rows))
        core = core | (np.abs(np.sqrt((i - 0.54 * rows) ** 2 + (j - 0.50 * cols) ** 2) - 0.30 * rows)
                       <= 0.02 * rows)
        eps = np.where(core, 10.0 * fd.EPS0, fd.EPS0).astype(np.float32)
        return eps, np.full((r1 - r0, cols), fd.MU0, np.float32)
    raise ValueError(kind)


def time_single(fd, rows, cols, steps, warmup, materials, device, boundary="mur"):
    """One whole-grid engine on one GPU.  Returns dict(wall_s, event_ms, launches, ...)."""
    import torch
    eng = fd.Engine(rows, cols, DT, DX, dtype=np.float32, device=device, boundary=boundary)
    eps, mu = make_materials(fd, materials, rows, cols)
    if eps is None:
        eng.set_materials()
    else:
        eng.set_materials(eps, mu, allow_uniform=(materials != "array"))
    del eps, mu
    if boundary == "pml":
        eng.set_pml()
    sr, sc = rows // 2, cols // 2
    eng.prepare(steps)          # launch-shape tuner (trial launches, state untouched): part of set-up
    eng.run(warmup, sr, sc, amplitudes(fd, 0, warmup)).sync()
    amps = amplitudes(fd, warmup, steps)
    l0 = eng.info(16), eng.info(17)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.timer_start()
    eng.run(steps, sr, sc, amps)
    ev_ms = eng.timer_stop()
    eng.sync()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    cyc = eng.cycle_steps if boundary == "mur" else min(8, eng.cycle_steps)
    res = dict(wall_s=wall, event_ms=ev_ms, pass_launches=eng.info(16) - l0[0],
               step_launches=eng.info(17) - l0[1], bpc=eng.bytes_per_cell_step, launch_steps=cyc,
               band_rows=eng.info(19), waves_per_strip=eng.info(20))
    # duration of the dominant kernel by itself: single full-length launches, each between its
    # own pair of HIP events on the engine's stream (what rocprofv3's kernel trace reports)
    if res["pass_launches"] and cyc and steps >= cyc:
        one = np.sort(eng.time_launches(48, cyc))
        res["launch_ms"] = float(np.mean(one[4:-4]))       # trimmed mean of back-to-back launches
    Ez, _, _ = eng.download()
    assert np.isfinite(Ez).all() and np.abs(Ez).max() > 0, "benchmark produced an empty field"
    eng.close()
    return res


def measured_traffic(rows, cols, materials, steps_per_launch):
    """HBM bytes per launch from the committed PMC profile of this exact configuration
    (profiles/r01_traffic.json), or None: counters cannot be read from inside the bench."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        e = t.get(f"{rows}x{cols}:{materials}")
        if e and abs(e["steps_per_launch"] - steps_per_launch) < 1e-9:
            return e["bytes_per_launch"]
    except Exception:
        pass
    return None


def roofline_block(cells, steps, r):
    """Per-launch algorithmic rate of the dominant kernel."""
    if r["pass_launches"]:
        launches = r["pass_launches"]
        name = ("k_bulk_split (temporally blocked, 16 steps per launch)" if r.get("launch_steps") == 16
                else "k_bulk (temporally blocked, up to 8 steps per launch)")
    else:
        launches, name = max(1, r["step_launches"] // 2), "k_update_h + k_update_e (one step)"
    region_ms = r["event_ms"] / launches                 # includes the gaps between launches
    if "launch_ms" in r:                                   # a full-length launch timed by itself
        ms, bytes_per_launch = r["launch_ms"], cells * r["launch_steps"] * r["bpc"]
    else:
        ms, bytes_per_launch = region_ms, cells * steps * r["bpc"] / launches
    ach = bytes_per_launch / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "kernel": name,
            "bytes_per_cell_step": r["bpc"], "launches": launches,
            "steps_per_launch": r.get("launch_steps") if "launch_ms" in r else round(steps / launches, 3),
            "launch_shape": {"band_rows": r.get("band_rows"), "waves_per_strip": r.get("waves_per_strip")},
            "avg_launch_ms": round(ms, 5), "avg_launch_ms_incl_gaps": round(region_ms, 5),
            "algorithmic_bytes_per_launch": int(bytes_per_launch)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--grid", type=int, default=0, help="rows (default 4096 per GPU)")
    ap.add_argument("--cols", type=int, default=0, help="columns (default 4096*gpus)")
    ap.add_argument("--materials", choices=["uniform", "array", "ring"], default="uniform")
    ap.add_argument("--boundary", choices=["mur", "pml"], default="mur",
                    help="mur = the reference's 5-px Mur frame; pml = the build-defined split-field PML")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import fdtd2d_amd as fd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                 f"--nproc-per-node {args.gpus}")
    if os.environ.get("FDTD2D_ONE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    rows = args.grid or SLAB_ROWS * world
    cols = args.cols or (args.grid if args.grid else SLAB_ROWS * world)
    cells = rows * cols

    if world == 1:
        r = time_single(fd, rows, cols, args.steps, args.warmup, args.materials, local, args.boundary)
        value = cells * args.steps / r["wall_s"] / 1e6
        res = {
            "metric": "Mcell-steps/s", "value": round(value, 1), "unit": "Mcell-steps/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(r["wall_s"] * 1e3 / args.steps, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{rows}x{cols} fp32 TE-mode, {args.materials} eps/mu, "
                                   + ("Mur-5" if args.boundary == "mur" else "split-field PML (40 cells)")
                                   + " boundary, ricker point source at the centre"
                                   + (" (BASELINE configs[1])" if (rows, cols, args.materials, args.boundary) ==
                                      (4096, 4096, "uniform", "mur") else ""),
                       "grid": [rows, cols], "materials": args.materials, "boundary": args.boundary},
            "roofline": roofline_block(cells, args.steps, r),
        }
        rl = res["roofline"]
        if args.boundary == "mur" and r["pass_launches"]:
            rl["traffic"] = measured_traffic(rows, cols, args.materials, r["launch_steps"])
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(rows)
        print(json.dumps(res))
        return

    # ---- N > 1: one process per GPU, row slabs, halo exchange over RCCL ---------------------
    import torch.distributed as dist
    from fdtd2d_amd.slab import SlabRunner
    # FDTD2D_DIST_BACKEND=gloo + FDTD2D_ONE_GPU=1: rehearsal of this code path with several
    # ranks on ONE GPU (halos staged through the host); never used for reported numbers.
    backend = os.environ.get("FDTD2D_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    sr, sc = rows // 2, cols // 2

    def make_runner(overlap):
        r_ = SlabRunner(rows, cols, DT, DX, dtype=np.float32, device=local, boundary=args.boundary,
                        overlap=overlap)
        lo, hi = r_.engine.stored_rows
        eps, mu = make_materials(fd, args.materials, rows, cols, lo, hi)
        r_.set_materials(eps, mu, allow_uniform=(args.materials != "array"))
        return r_

    exchange_mode = "overlapped"
    runner = make_runner(True)
    try:
        runner.run(max(args.warmup, 32), sr, sc, amplitudes(fd, 0, max(args.warmup, 32)))
        torch.cuda.synchronize()
    except Exception as exc:      # deterministic on every rank: fall back to the plain cycle
        print(f"[rank {rank}] overlapped exchange failed ({exc!r}); using the plain exchange cycle",
              file=sys.stderr, flush=True)
        exchange_mode = "plain"
        runner.close()
        runner = make_runner(False)
        runner.run(max(args.warmup, 32), sr, sc, amplitudes(fd, 0, max(args.warmup, 32)))
    amps = amplitudes(fd, args.warmup, args.steps)
    cycle = min(runner.halo, runner.engine.cycle_steps or runner.halo)   # steps per exchange
    runner.prepare(args.steps)      # kernels of the last, shorter cycle: part of set-up
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.run(args.steps, sr, sc, amps)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    wall = torch.tensor([time.perf_counter() - t0], device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    wall = float(wall.item())
    ok = runner.sanity()
    slab = runner.engine.nrows
    runner.close()
    if rank == 0:
        single = time_single(fd, slab, cols, min(args.steps, 96), 16, args.materials, local, args.boundary)
        single_v = slab * cols * min(args.steps, 96) / single["wall_s"] / 1e6
        value = cells * args.steps / wall / 1e6
        bpc = single["bpc"]
        ach = value * 1e6 * bpc / 1e9
        res = {
            "metric": "Mcell-steps/s", "value": round(value, 1), "unit": "Mcell-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall * 1e3 / args.steps, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{rows}x{cols} fp32 TE-mode, {args.materials} eps/mu, {args.boundary} "
                                   f"boundary, {world} row slabs of {slab} rows, halo {cycle} rows of "
                                   f"Ez/Hx/Hy every {cycle} steps over {backend} send/recv",
                       "grid": [rows, cols], "materials": args.materials,
                       "per_gpu_slab": [slab, cols], "fields_finite": bool(ok),
                       "exchange": exchange_mode},
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS * world,
                         "unit": "GB/s", "frac": round(ach / (HBM_PEAK_GBS * world), 4),
                         "traffic": None, "kernel": "k_bulk_split (16-step passes), whole job (all ranks)" if cycle == 16 else "k_bulk_split / k_pass_pml (8-step passes), whole job (all ranks)",
                         "bytes_per_cell_step": bpc},
            "weak_scaling": {"single_gpu_same_slab": round(single_v, 1),
                             "efficiency": round(value / (world * single_v), 4)},
        }
        print(json.dumps(res))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
