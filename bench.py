#!/usr/bin/env python3
"""Benchmark of the FDTD hot path on MI355X.  Prints ONE JSON line (rank 0).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--grid R [--cols C]]
                  [--materials uniform|array|ring] [--boundary mur|pml]
                  [--pmc live|profile|off] [--no-secondary] [--no-cpu-baseline]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full leapfrog step (H half-step, E half-step with the boundary frame, point
source) over the whole grid -- fdtd.py:31-34 of the reference.

Workloads (all synthetic: zero fields, ricker source at the grid centre, fp32):
  N = 1  headline: 16384 x 16384, uniform eps/mu, Mur-5 frame -- the grid BASELINE.json's
         north_star quotes its 1-GPU target on.  The same JSON line carries, under "secondary",
         BASELINE configs[1] (4096^2 uniform) and configs[2] (8192^2 ring-resonator eps map).
  N > 1  row slabs of 4096 rows per GPU, columns 4096*N: configs[3] (16384^2 on 4, Mur-5) and
         configs[4] (32768^2 on 8, PML) and their 2-GPU sibling; one process per GPU, halo
         exchange over torch.distributed (RCCL).  The same slab shape is also timed on one GPU
         alone ("single_gpu_same_slab").

Timing: inputs resident in HBM; W untimed warm-up steps; barrier + device sync; K steps;
device sync + barrier; max over ranks.  value = cells * K / time.

roofline (dominant kernel = the temporally blocked pass, one launch = `steps_per_launch` steps):
  achieved / peak / frac   REAL HBM bytes per launch (traffic) / launch duration, against 8.0 TB/s.
                           A pass keeps 16 time levels on chip, so the algorithmic byte count of the
                           step-by-step formulation is not what the kernel moves; the fraction that
                           still bounds it is the real one.
  traffic                  HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x 2 per the
                           gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE exact), collected by
                           THIS run in child processes (traffic_source says so) or, if that is not
                           possible, taken from the committed profile of the same kernel, grid and
                           launch shape (profiles/r02_traffic.json).
  overfetch                traffic / (one read + one write of every field = 24 B x cells (+4 B per
                           coefficient array)): 1.0 would be a pass without overlap re-reads.
  valu_frac                VALU wave-instructions per launch (SQ_INSTS_VALU) x 2 cycles / (1024 SIMDs
                           x launch duration x 2.4 GHz): share of the wave64 issue peak.
  algorithmic              SURVEY.md section 8 M2's figure (24 B per cell-step, +4 per coefficient array)
                           x cells x steps per launch / launch duration, and its ratio to 8.0 TB/s
                           (> 1 means: fewer real bytes than a one-step-per-pass kernel must move).
  avg_launch_ms            trimmed mean of 48 back-to-back full-length launches, each between its own
                           pair of HIP events on the engine's stream (what rocprofv3's kernel trace
                           reports per dispatch); steady_state_value = the rate of such launches.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DT, DX, FC = 5e-14, 1e-4, 30e9
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec); 6.29 TB/s measured copy
SIMDS, PEAK_GHZ, VALU_ISSUE_CYCLES = 1024, 2.4, 2.0      # 256 CUs x 4 SIMD-32; wave64 VALU = 2 cycles
SLAB_ROWS = 4096
TRAFFIC_PROFILE = os.path.join(ROOT, "profiles", "r02_traffic.json")


def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count()


def cpu_baseline(grid: int, budget_s: float = 12.0):
    """The NumPy oracle -- expression for expression the structure of the reference's NumPy
    code, single core like it -- on a bounded sample of the N=1 workload; plus the OpenMP C
    oracle on all host cores for context.  Baseline, not target."""
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
    model, ncpu = cpu_info()
    out = {"value": round(g * g * n / el / 1e6, 2), "unit": "Mcell-steps/s", "cores": 1,
           "kind": "port", "cpu_model": model, "os_cpu_count": ncpu,
           "sample": f"NumPy oracle (oracle/fdtd_numpy.py), {g}x{g} fp32 vacuum block of the workload, "
                     f"{n} steps in {el:.1f}s"}
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


def make_engine(fd, rows, cols, materials, device, boundary, shape=None, autotune=True):
    eng = fd.Engine(rows, cols, DT, DX, dtype=np.float32, device=device, boundary=boundary)
    eps, mu = make_materials(fd, materials, rows, cols)
    if eps is None:
        eng.set_materials()
    else:
        eng.set_materials(eps, mu, allow_uniform=(materials != "array"))
    del eps, mu
    if boundary == "pml":
        eng.set_pml()
    if not autotune:
        eng.set_option(autotune=False)
    if shape and shape[0]:
        eng.set_option(long_shape=shape)
    return eng


def time_single(fd, rows, cols, steps, warmup, materials, device, boundary="mur", shape=None, autotune=True):
    """One whole-grid engine on one GPU.  Returns dict(wall_s, event_ms, launches, ...)."""
    import torch
    eng = make_engine(fd, rows, cols, materials, device, boundary, shape, autotune)
    sr, sc = rows // 2, cols // 2
    eng.prepare(steps, sr, sc)  # launch-shape tuner (trial launches, state untouched): part of set-up
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
    cyc = eng.cycle_steps
    res = dict(wall_s=wall, event_ms=ev_ms, pass_launches=eng.info(16) - l0[0],
               step_launches=eng.info(17) - l0[1], bpc=eng.bytes_per_cell_step, launch_steps=cyc,
               band_rows=eng.info(19), waves_per_strip=eng.info(20), edge_rows=eng.info(21))
    # duration of the dominant kernel by itself: single full-length launches, each between its
    # own pair of HIP events on the engine's stream (what rocprofv3's kernel trace reports)
    if cyc:
        eng.run(cyc).sync()      # (also makes the full-length shape the "last" one when steps < cyc)
        res["band_rows"], res["waves_per_strip"], res["edge_rows"] = eng.last_shape
        one = np.sort(eng.time_launches(48, cyc))
        res["launch_ms"] = float(np.mean(one[4:-4]))       # trimmed mean of back-to-back launches
    Ez, _, _ = eng.download()
    assert np.isfinite(Ez).all() and np.abs(Ez).max() > 0, "benchmark produced an empty field"
    eng.close()
    return res


# ---- HBM traffic of the pass kernel, measured by this run -----------------------------------------

def pmc_child(args):
    """`bench.py --pmc-child`: a short steady-state sequence of full-length passes, run under
    `rocprofv3 --pmc ...` by measure_traffic().  Prints the launch shape it used."""
    import fdtd2d_amd as fd
    shape = (args.band_rows, args.waves, args.edge_rows) if args.band_rows else None
    eng = make_engine(fd, args.grid, args.cols, args.materials, 0, args.boundary, shape)
    cyc = eng.cycle_steps
    n = cyc * 12
    eng.prepare(n, args.grid // 2, args.cols // 2)
    eng.run(n, args.grid // 2, args.cols // 2, amplitudes(fd, 0, n)).sync()
    print(json.dumps({"pmc_child": True, "shape": list(eng.last_shape), "cycle": cyc}), flush=True)
    eng.close()


def _pmc_run(counters, child_args, timeout=240):
    """One rocprofv3 PMC pass over a child bench process -> ({kernel: {counter: [values]}}, child info)."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        raise RuntimeError("rocprofv3 not on PATH")
    out = tempfile.mkdtemp(prefix="fdtd2d_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", *child_args]
        p = subprocess.run(cmd, cwd=out, capture_output=True, text=True, timeout=timeout)
        info = None
        for line in p.stdout.splitlines():
            if line.startswith("{") and "pmc_child" in line:
                info = json.loads(line)
        acc = {}
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                acc.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        if info is None or not acc:
            raise RuntimeError(f"rocprofv3 child produced no counters (exit code {p.returncode}): "
                               f"{(p.stderr or p.stdout)[-300:]}")
        return acc, info
    finally:
        shutil.rmtree(out, ignore_errors=True)


def _dominant(acc, counter):
    """(name of the pass kernel with the most dispatches, steady-state `counter` per PASS): a pass may be
    several kernels launched once each -- the PML pair k_bulk_split + k_bulk_split_pml, k_zone beside the
    bulk -- whose counters add up."""
    cands = {k: d[counter] for k, d in acc.items()
             if counter in d and any(t in k for t in ("k_bulk", "k_pass", "k_zone"))}
    if not cands:
        raise RuntimeError(f"no pass kernel with {counter} among {list(acc)[:4]}")
    main = max((k for k in cands if "k_zone" not in k), key=lambda k: len(cands[k]), default=None)
    if main is None:
        raise RuntimeError(f"no pass kernel with {counter} among {list(acc)[:4]}")
    n = len(cands[main])
    base = lambda k: k.split("<")[0].split("::")[-1].strip()
    total = 0.0
    for k, v in cands.items():
        # launched (about) once per pass, and not another instantiation of the main template (those are the
        # tuner's trial variants, e.g. 8 waves per strip)
        if 2 * len(v) >= n and (k == main or base(k) != base(main)):
            w = v[len(v) // 2:]               # second half: steady state, shape tuned
            total += sum(w) / len(w)
    return main, total


def _tune_run(child_args, timeout=240):
    """The same child WITHOUT the profiler: the launch-shape tuner times trial launches, and under counter
    collection (serialised dispatches, per-dispatch overhead) it picked shapes up to 13 % slower than on a
    quiet GPU (16384^2: 208-row bands 1.57 ms where 315 / 139 gives 1.38 ms)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", *child_args],
                       capture_output=True, text=True, timeout=timeout)
    for line in p.stdout.splitlines():
        if line.startswith("{") and "pmc_child" in line:
            return json.loads(line)
    raise RuntimeError(f"tuning child failed (exit code {p.returncode}): {(p.stderr or p.stdout)[-300:]}")


def measure_traffic(rows, cols, materials, boundary):
    """A tuning child on the quiet GPU fixes the launch shape; two PMC passes (FETCH_SIZE does not fit beside
    WRITE_SIZE: MI355X_MICROARCH.md, PMC slots) and the caller then run exactly that shape."""
    base = ["--grid", str(rows), "--cols", str(cols), "--materials", materials, "--boundary", boundary]
    tuned = _tune_run(base)
    base += ["--band-rows", str(tuned["shape"][0]), "--waves", str(tuned["shape"][1]),
             "--edge-rows", str(tuned["shape"][2])]
    acc, info = _pmc_run(["FETCH_SIZE", "SQ_INSTS_VALU"], base)
    kernel, fetch_kib = _dominant(acc, "FETCH_SIZE")
    valu = None
    try:
        valu = _dominant(acc, "SQ_INSTS_VALU")[1]
    except RuntimeError:
        pass
    shape = tuple(int(v) for v in info["shape"])
    acc2, _ = _pmc_run(["WRITE_SIZE"], base)
    _, write_kib = _dominant(acc2, "WRITE_SIZE")
    rd, wr = 2.0 * fetch_kib * 1024, write_kib * 1024      # gfx950: FETCH_SIZE tallies 128-B requests as 64 B
    return {"bytes_per_launch": int(rd + wr), "read": int(rd), "write": int(wr), "valu_insts": valu,
            "kernel": kernel.split("(")[0].replace("void ", ""), "shape": shape, "steps_per_launch": info["cycle"],
            "source": "rocprofv3 --pmc FETCH_SIZE (x2) / WRITE_SIZE, separate passes, collected by this run"}


def profile_traffic(rows, cols, materials, boundary, steps_per_launch):
    """Fallback: the committed PMC figure of the same grid / materials / boundary / pass length."""
    try:
        t = json.load(open(TRAFFIC_PROFILE))
        e = t.get(f"{rows}x{cols}:{materials}:{boundary}")
        if e and e["steps_per_launch"] == steps_per_launch:
            e = dict(e)
            e["source"] = f"committed profile {e.get('source', TRAFFIC_PROFILE)} (not measured in this run)"
            e["shape"] = tuple(e.get("shape", (0, 0)))
            return e
    except Exception:
        pass
    return None


def roofline_block(cells, steps, r, traffic):
    """Roofline of the dominant kernel (see the module docstring for every field)."""
    n_arr = (r["bpc"] - 24) // 4
    if r["pass_launches"] or "launch_ms" in r:
        launches = max(1, r["pass_launches"])
        name = ("k_bulk_split (temporally blocked, level-split)" if r.get("waves_per_strip", 1) > 1
                else "k_bulk / k_pass_pml (temporally blocked, one wave per strip)")
    else:
        launches, name = max(1, r["step_launches"] // 2), "k_update_h + k_update_e (one step)"
    region_ms = r["event_ms"] / launches                 # includes the gaps between launches
    if "launch_ms" in r:                                   # a full-length launch timed by itself
        ms, spl = r["launch_ms"], r["launch_steps"]
    else:
        ms, spl = region_ms, steps / launches
    alg_bytes = cells * spl * r["bpc"]
    alg_rate = alg_bytes / (ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
           "traffic": None, "traffic_source": None, "overfetch": None, "valu_frac": None,
           "kernel": name, "steps_per_launch": spl,
           "launch_shape": {"band_rows": r.get("band_rows"), "waves_per_strip": r.get("waves_per_strip"),
                            "edge_strip_band_rows": r.get("edge_rows")},
           "avg_launch_ms": round(ms, 5), "avg_launch_ms_incl_gaps": round(region_ms, 5),
           "steady_state_value": round(cells * spl / (ms * 1e-3) / 1e6, 1),
           "algorithmic": {"bytes_per_cell_step": r["bpc"], "bytes_per_launch": int(alg_bytes),
                           "rate_GBps": round(alg_rate, 1), "x_peak": round(alg_rate / HBM_PEAK_GBS, 3)}}
    if traffic and traffic.get("steps_per_launch") == spl:
        real = traffic["bytes_per_launch"] / (ms * 1e-3) / 1e9
        out.update(achieved=round(real, 1), frac=round(real / HBM_PEAK_GBS, 4),
                   traffic=traffic["bytes_per_launch"], traffic_source=traffic["source"],
                   traffic_read_write=[traffic.get("read"), traffic.get("write")],
                   overfetch=round(traffic["bytes_per_launch"] / (cells * (24 + 4 * n_arr)), 3))
        if traffic.get("kernel"):
            out["kernel"] = traffic["kernel"] + f" ({spl} steps per launch)"
        if traffic.get("valu_insts"):
            out["valu_frac"] = round(traffic["valu_insts"] * VALU_ISSUE_CYCLES /
                                     (SIMDS * ms * 1e-3 * PEAK_GHZ * 1e9), 4)
            out["valu_insts_per_launch"] = int(traffic["valu_insts"])
    return out


def single_record(fd, rows, cols, steps, warmup, materials, boundary, device, traffic, pmc, note="", shape=None,
                  autotune=True):
    """Timing of one whole-grid configuration + its roofline block.  traffic: the live PMC result
    (dict), an error string if the live measurement failed, or None."""
    err = traffic if isinstance(traffic, str) else None
    traffic = None if err else traffic
    r = time_single(fd, rows, cols, steps, warmup, materials, device, boundary,
                    traffic["shape"] if traffic else shape, autotune)
    if traffic is None and pmc != "off":
        traffic = profile_traffic(rows, cols, materials, boundary, r["launch_steps"])
    cells = rows * cols
    rec = {"value": round(cells * steps / r["wall_s"] / 1e6, 1), "unit": "Mcell-steps/s", "steps": steps,
           "warmup": warmup, "ms_per_step": round(r["wall_s"] * 1e3 / steps, 5),
           "config": {"workload": f"{rows}x{cols} fp32 TE-mode, {materials} eps/mu, "
                                  + ("Mur-5" if boundary == "mur" else "split-field PML (40 cells)")
                                  + " boundary, ricker point source at the centre" + note,
                      "grid": [rows, cols], "materials": materials, "boundary": boundary},
           "roofline": roofline_block(cells, steps, r, traffic)}
    if err:
        rec["roofline"]["traffic_note"] = err
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--grid", type=int, default=0, help="rows (default 16384 on one GPU, 4096 per GPU otherwise)")
    ap.add_argument("--cols", type=int, default=0, help="columns (default = rows on one GPU, 4096*gpus otherwise)")
    ap.add_argument("--materials", choices=["uniform", "array", "ring"], default="uniform")
    ap.add_argument("--boundary", choices=["mur", "pml"], default=None,
                    help="mur = the reference's 5-px Mur frame; pml = the build-defined split-field PML "
                         "(default: mur, and pml for --gpus 8 = BASELINE configs[4])")
    ap.add_argument("--pmc", choices=["live", "profile", "off"], default="live",
                    help="roofline.traffic: measured by this run through rocprofv3 child processes, taken "
                         "from the committed profile, or omitted")
    ap.add_argument("--exchange", choices=["overlapped", "plain"], default="overlapped")
    ap.add_argument("--loop", choices=["auto", "c", "python"], default="auto",
                    help="N > 1: the run loop in C with the library's RCCL transport (auto on nccl) or the "
                         "Python-sequenced cycle over torch.distributed")
    ap.add_argument("--no-secondary", action="store_true", help="N=1: skip the configs[1] / configs[2] records")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-autotune", action="store_true",
                    help="fixed launch-shape rules (for profiler runs: the tuner's trial launches would be "
                         "averaged into the per-kernel statistics); combine with --band-rows/--waves/--edge-rows")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--band-rows", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--waves", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--edge-rows", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.pmc_child:
        args.cols = args.cols or args.grid
        args.boundary = args.boundary or "mur"
        return pmc_child(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                 f"--nproc-per-node {args.gpus}")
    if os.environ.get("FDTD2D_ONE_GPU"):
        local = 0
    boundary = args.boundary or ("pml" if world == 8 else "mur")

    if world == 1:
        rows = args.grid or 16384
        cols = args.cols or rows
        default_cfg = (rows, cols, args.materials, boundary) == (16384, 16384, "uniform", "mur")
        recs = [(rows, cols, args.steps, args.warmup, args.materials, boundary,
                 " (the grid of BASELINE's 1-GPU target)" if default_cfg else "")]
        if default_cfg and not args.no_secondary:
            recs += [(4096, 4096, 400, 40, "uniform", "mur", " (BASELINE configs[1])"),
                     (8192, 8192, 160, 32, "ring", "mur", " (BASELINE configs[2] geometry)")]
        # HBM traffic first: the rocprofv3 child processes must start before THIS process has
        # initialised the GPU (an exec from a GPU-initialised process is refused on this pool)
        traffic = []
        for (r_, c_, st, wu, mat, bnd, note) in recs:
            t = None
            if args.pmc == "live":
                try:
                    t = measure_traffic(r_, c_, mat, bnd)
                except Exception as exc:
                    t = f"live PMC failed ({exc}); fell back to the committed profile"
            traffic.append(t)
        import torch
        import fdtd2d_amd as fd
        torch.cuda.set_device(local)
        forced = (args.band_rows, args.waves, args.edge_rows) if args.band_rows else None
        out = [single_record(fd, r_, c_, st, wu, mat, bnd, local, t, args.pmc, note, forced if k == 0 else None,
                             not args.no_autotune)
               for k, ((r_, c_, st, wu, mat, bnd, note), t) in enumerate(zip(recs, traffic))]
        head = out[0]
        res = {"metric": "Mcell-steps/s", "value": head["value"], "unit": "Mcell-steps/s",
               "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic", "config": head["config"], "roofline": head["roofline"]}
        if len(out) > 1:
            res["secondary"] = out[1:]
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(rows)
        print(json.dumps(res))
        return

    # ---- N > 1: one process per GPU, row slabs, halo exchange over RCCL ---------------------
    import torch
    import torch.distributed as dist
    import fdtd2d_amd as fd
    from fdtd2d_amd.slab import SlabRunner
    torch.cuda.set_device(local)
    rows = args.grid or SLAB_ROWS * world
    cols = args.cols or (args.grid if args.grid else SLAB_ROWS * world)
    cells = rows * cols
    # FDTD2D_DIST_BACKEND=gloo + FDTD2D_ONE_GPU=1: rehearsal of this code path with several
    # ranks on ONE GPU (halos staged through the host); never used for reported numbers.
    backend = os.environ.get("FDTD2D_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    sr, sc = rows // 2, cols // 2
    try:
        # No fallback between exchange modes: a failure is printed by the rank that saw it and ends
        # the job (a retry in the same process could match stale messages of the failed attempt).
        runner = SlabRunner(rows, cols, DT, DX, dtype=np.float32, device=local, boundary=boundary,
                            overlap=(args.exchange == "overlapped"), loop=args.loop)
        lo, hi = runner.engine.stored_rows
        eps, mu = make_materials(fd, args.materials, rows, cols, lo, hi)
        runner.set_materials(eps, mu, allow_uniform=(args.materials != "array"))
        del eps, mu
        wu = max(args.warmup, 2 * max(runner.cycle, 8))
        runner.run(wu, sr, sc, amplitudes(fd, 0, wu))
        amps = amplitudes(fd, wu, args.steps)
        cycle = runner.cycle
        runner.prepare(args.steps)      # kernels of the last, shorter cycle: part of set-up
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.run(args.steps, sr, sc, amps)
        host_s = time.perf_counter() - t0           # host time to ENQUEUE the whole run
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        wall = torch.tensor([time.perf_counter() - t0], device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(wall, op=dist.ReduceOp.MAX)
        wall = float(wall.item())
        ok = runner.sanity()
        slab = runner.engine.nrows
        exchange_mode = "overlapped" if runner.overlap and rows >= 2 * (2 * cycle + 6) else "plain"
        loop_used = runner.loop + (" (library RCCL transport)" if runner.loop == "c" and backend == "nccl" else "")
        runner.close()
    except Exception:
        print(f"[rank {rank}] slab run failed:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
        os._exit(3)
    if rank == 0:
        single = time_single(fd, slab, cols, min(args.steps, 96), 16, args.materials, local, boundary)
        single_v = slab * cols * min(args.steps, 96) / single["wall_s"] / 1e6
        value = cells * args.steps / wall / 1e6
        bpc = single["bpc"]
        alg = value * 1e6 * bpc / 1e9
        ncyc = max(1, -(-args.steps // max(cycle, 1)))
        res = {
            "metric": "Mcell-steps/s", "value": round(value, 1), "unit": "Mcell-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": wu,
            "ms_per_step": round(wall * 1e3 / args.steps, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{rows}x{cols} fp32 TE-mode, {args.materials} eps/mu, {boundary} "
                                   f"boundary, {world} row slabs of {slab} rows, halo {runner.halo} rows of "
                                   f"every field every {cycle} steps over {backend} send/recv",
                       "grid": [rows, cols], "materials": args.materials, "boundary": boundary,
                       "per_gpu_slab": [slab, cols], "fields_finite": bool(ok),
                       "exchange": exchange_mode, "cycle_steps": cycle, "loop": loop_used,
                       "host_us_per_cycle": round(host_s * 1e6 / ncyc, 1)},
            "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": None, "traffic": None,
                         "kernel": "pass kernels of all ranks (whole job); per-kernel figures: the N=1 line",
                         "algorithmic": {"bytes_per_cell_step": bpc, "rate_GBps": round(alg, 1),
                                         "x_peak": round(alg / (HBM_PEAK_GBS * world), 3)}},
            "weak_scaling": {"single_gpu_same_slab": round(single_v, 1),
                             "efficiency": round(value / (world * single_v), 4)},
        }
        print(json.dumps(res))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
