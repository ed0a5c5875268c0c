#!/usr/bin/env python3
"""Benchmark of the FDTD hot path on MI355X.  Prints ONE JSON line (rank 0).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R] [--grid R [--cols C]]
                  [--materials uniform|array|ring] [--boundary mur|pml] [--dtype f32|f64]
                  [--pmc live|off] [--no-secondary] [--no-cpu-baseline]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full leapfrog step (H half-step, E half-step with the boundary frame, point
source) over the whole grid -- fdtd.py:31-34 of the reference.

Workloads (all synthetic: zero fields, ricker source at the grid centre):
  N = 1  headline: 16384 x 16384 fp32, uniform eps/mu, Mur-5 frame -- the grid BASELINE.json's
         north_star quotes its 1-GPU target on.  The same JSON line carries, under "secondary",
         BASELINE configs[1] (4096^2 uniform), configs[2] (8192^2 ring-resonator eps map), the
         headline grid with eps AND mu as arrays (32 B per cell-step, SURVEY.md M2's byte count), one rank's
         4096 x 32768 slab of configs[4] with the PML, and the reference's own arithmetic type, float64, at
         4096^2 and 16384^2.
  N > 1  row slabs of 4096 rows per GPU, columns 4096*N: configs[3] (16384^2 on 4, Mur-5) and
         configs[4] (32768^2 on 8, PML) and their 2-GPU sibling; one process per GPU, halo exchange
         over RCCL.  After the timed run every rank CHECKS the rows next to its cuts against a
         single-engine run of a sub-grid around the cut (cut_bands_identical; exit code 4 when false).

Timing (N = 1): inputs resident in HBM; W untimed warm-up steps; then R repetitions of: device sync,
K steps, device sync.  value = cells * K / MEDIAN wall time of the repetitions (min / max beside it);
every repetition also carries the HIP-event time of its K steps on the engine's stream.

roofline = the kernels that produced `value` (the passes of ONE run of K steps; with the driver's K = 20 at
16384^2 that is the 20-step pair k_bulk_split<20> + k_zone<wide>):
  achieved / peak / frac   REAL HBM bytes of one K-step run (traffic) / median HIP-event time of the run,
                           against 8.0 TB/s.  A pass keeps 16-20 time levels on chip, so the algorithmic
                           byte count of the step-by-step formulation is not what the kernels move; the
                           fraction that still bounds them is the real one.
  traffic                  HBM bytes per LAUNCH (= per pass; traffic_per_run / passes_per_run, like avg_launch_ms) from rocprofv3 PMC passes (FETCH_SIZE x 2 per the
                           gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE exact), collected by THIS
                           run in child processes running the same K-step plan on the same launch shapes.
  overfetch                traffic / (one read + one write of every field per pass = 24 B x cells (+4 B per
                           coefficient array) x passes per run): 1.0 would be passes without overlap re-reads.
  valu_frac                VALU wave-instructions per run (SQ_INSTS_VALU) x 2 cycles / (1024 SIMDs x run
                           time x 2.4 GHz): share of the wave64 issue peak.
  algorithmic              SURVEY.md section 8 M2's figure (24 B per cell-step, +4 per coefficient array)
                           x cells x K / run time, and its ratio to 8.0 TB/s (> 1 means: fewer real bytes
                           than a one-step-per-pass kernel must move).
  frac_of_copy_rate        achieved / what a plain 16-byte-per-lane copy of the same field arrays reaches on this box in
                           this run (copy_kernel_GBps_before_after): the practical ceiling beside the data sheet's.
  steady_state             the full-length pass kernel (16 steps) by itself: trimmed mean of
                           48 back-to-back launches, each between its own pair of HIP events on the engine's
                           stream (what rocprofv3's kernel trace reports per dispatch), with its own traffic.
gpu_state = clocks per XCD, memory clock, socket power, temperature and the throttle-residency counters read
through amdsmi before, DURING (sampling thread, ~1 kHz) and after the timed region.
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
import threading
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DT, DX, FC = 5e-14, 1e-4, 30e9
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec); 6.29 TB/s measured copy
SIMDS, PEAK_GHZ, VALU_ISSUE_CYCLES = 1024, 2.4, 2.0      # 256 CUs x 4 SIMD-32; wave64 VALU = 2 cycles
SLAB_ROWS = 4096
NP_DTYPE = {"f32": np.float32, "f64": np.float64}
SHAPE_KEYS = ("band_rows", "waves_per_level_group", "edge_strip_band_rows", "waves_side_by_side", "xcd_map",
              "filler_band_rows", "filler_bands_per_strip", "zone_tiles_fused")
ENGINE_OPTS = {}     # --opt name=value: Engine.set_option() knobs for A/B experiments (speed only, same results)


# ---- GPU clocks / power / throttling (amdsmi) -------------------------------------------------------

class GpuState:
    """Reads the card's gpu_metrics through amdsmi (0.6 ms per call on the test box).  sample() runs a
    thread that polls while the timed region executes; summary() condenses it."""
    ACC = ("ppt_residency_acc", "prochot_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc",
           "hbm_thm_residency_acc")

    def __init__(self, device=0):
        self.h, self.err = None, None
        try:
            import amdsmi
            self.smi = amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self.h = hs[min(device, len(hs) - 1)]
            self.partition = (str(amdsmi.amdsmi_get_gpu_compute_partition(self.h)),
                              str(amdsmi.amdsmi_get_gpu_memory_partition(self.h)))
            # which physical device this is (the same launch takes 1.28 ms on some devices and 1.48 ms on others at
            # the same clocks and the same copy rate: DESIGN.md section 5.2) and the firmware it runs
            self.device = {}
            for name, keys in (("amdsmi_get_gpu_asic_info", ("market_name", "device_id", "rev_id", "asic_serial", "oam_id")),
                               ("amdsmi_get_gpu_vbios_info", ("version", "build_date", "part_number")),
                               ("amdsmi_get_gpu_device_bdf", None)):
                try:
                    v = getattr(amdsmi, name)(self.h)
                    if keys is None:
                        self.device["bdf"] = str(v)
                    else:
                        self.device.update({k: str(v[k]) for k in keys if k in v})
                except Exception:
                    pass
        except Exception as exc:     # no amdsmi / no permission: the block says so
            self.err = f"{type(exc).__name__}: {exc}"

    def read(self):
        if self.h is None:
            return None
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        num = lambda v: v if isinstance(v, (int, float)) else None
        out = {"gfx_mhz": [num(v) for v in m.get("current_gfxclks", [])][:8], "uclk_mhz": num(m.get("current_uclk")),
               "socclk_mhz": num(m.get("current_socclk")),
               "socket_w": num(m.get("current_socket_power")), "hotspot_c": num(m.get("temperature_hotspot")),
               "mem_c": num(m.get("temperature_mem")), "gfx_busy": num(m.get("average_gfx_activity")),
               "acc_n": num(m.get("accumulation_counter"))}
        for k in self.ACC:
            out[k] = num(m.get(k))
        return out

    def sample(self, period_s=0.001):
        self._rows, self._stop = [], threading.Event()

        def loop():
            while not self._stop.is_set():
                try:
                    self._rows.append(self.read())
                except Exception:
                    pass
                time.sleep(period_s)
        self._thr = threading.Thread(target=loop, daemon=True)
        if self.h is not None:
            self._thr.start()
        return self

    def summary(self):
        if self.h is None:
            return {"error": self.err}
        self._stop.set()
        self._thr.join()
        rows = [r for r in self._rows if r]
        if not rows:
            return {"samples": 0}
        # samples taken while the shader clock is up (the region also holds host gaps between repetitions)
        busy = [r for r in rows if r["gfx_mhz"] and min(v for v in r["gfx_mhz"] if v is not None) >= 1000] or rows
        gfx = np.array([[v or 0 for v in r["gfx_mhz"]] for r in busy], dtype=np.float64)
        col = lambda k, rs=busy: [r[k] for r in rs if r.get(k) is not None]
        med = lambda v: float(np.median(v)) if len(v) else None
        out = {"samples": len(rows), "busy_samples": len(busy),
               "gfx_mhz": {"min": float(gfx.min()), "median": float(np.median(gfx)), "max": float(gfx.max()),
                           "per_xcd_median": [float(x) for x in np.median(gfx, axis=0)]},
               "uclk_mhz": med(col("uclk_mhz")), "socket_w": {"median": med(col("socket_w")),
                                                             "max": max(col("socket_w"), default=None)},
               "hotspot_c_max": max(col("hotspot_c"), default=None), "mem_c_max": max(col("mem_c"), default=None)}
        first, last = rows[0], rows[-1]
        if first.get("acc_n") is not None and last.get("acc_n") is not None and last["acc_n"] > first["acc_n"]:
            dn = last["acc_n"] - first["acc_n"]
            out["throttle_residency"] = {k.replace("_residency_acc", ""): round((last[k] - first[k]) / dn, 4)
                                         for k in self.ACC if first.get(k) is not None and last.get(k) is not None}
        return out


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


def cpu_baseline(grid: int, budget_s: float = 10.0):
    """The NumPy oracle -- expression for expression the structure of the reference's NumPy code, single core
    like it -- on bounded samples: a 4096^2 fp32 block of the N=1 workload, and BASELINE configs[0] itself
    (256^2, 500 steps, float64 = the reference's default, and float32); plus the OpenMP C oracle on all host
    cores for context.  Baseline, not target."""
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
    cfg1 = {}
    for name, dt_ in (("f64", np.float64), ("f32", np.float32)):
        Ez, Hx, Hy = onp.grid_zeros(256, 256, dt_)
        e1, m1 = onp.vacuum_materials(256, 256, dt_)
        t0 = time.perf_counter()
        onp.leapfrog(Ez, Hx, Hy, e1, m1, DT, DX, 500, 128, 128)
        el = time.perf_counter() - t0
        cfg1[name] = {"value": round(256 * 256 * 500 / el / 1e6, 2), "ms_per_step": round(el * 2, 4),
                      "max_abs_Ez": float(np.abs(Ez).max())}
    out["config0_256x256_500_steps"] = dict(cfg1, cores=1, sample="NumPy oracle, BASELINE configs[0] in full")
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


def make_materials(fd, kind, rows, cols, r0=0, r1=None, dtype=np.float32):
    """eps, mu for global rows [r0, r1) -- None, None for the uniform case."""
    r1 = rows if r1 is None else r1
    if kind == "uniform":
        return None, None
    if kind == "array":            # uniform values handed over as full arrays, detection off
        return (np.full((r1 - r0, cols), fd.EPS0, dtype), np.full((r1 - r0, cols), fd.MU0, dtype))
    if kind == "ring":             # BASELINE configs[2] geometry
        i = np.arange(r0, r1, dtype=np.float64)[:, None]
        j = np.arange(cols, dtype=np.float64)[None, :]
        core = (i >= np.floor(0.18 * rows)) & (i < np.floor(0.22 * rows))
        core = core | (np.abs(np.sqrt((i - 0.54 * rows) ** 2 + (j - 0.50 * cols) ** 2) - 0.30 * rows)
                       <= 0.02 * rows)
        eps = np.where(core, 10.0 * fd.EPS0, fd.EPS0).astype(dtype)
        return eps, np.full((r1 - r0, cols), fd.MU0, dtype)
    raise ValueError(kind)


CLOCK_WARMUP_MS = 30.0


def clock_warmup(eng, cyc=0, target_ms=CLOCK_WARMUP_MS):
    """Untimed, right before a timed region: keep the GPU busy for ~30 ms.  An idle chip needs 10-20 ms of work before it
    holds its clock (the repetitions of a 1.7 ms run kept getting shorter from the 2nd to the 11th, and the set-up steps
    before them -- an amdsmi sample, a barrier -- are idle gaps; profiles/r03_clock_vs_launch.txt).  With cyc > 0: full-length
    passes on the engine's own fields (a whole-grid engine: the synthetic state simply moves on); else the library's copy
    kernel, which leaves the state alone (slab engines, whose halos a pass without an exchange would invalidate).  Returns
    the milliseconds spent."""
    t0 = time.perf_counter()
    for _ in range(64):
        if (time.perf_counter() - t0) * 1e3 >= target_ms:
            break
        if cyc > 0:
            eng.time_launches(8, cyc)
        else:
            eng.measure_copy(8)
    return round((time.perf_counter() - t0) * 1e3, 1)


def make_engine(fd, rows, cols, materials, device, boundary, shapes=None, autotune=True, dtype=np.float32):
    eng = fd.Engine(rows, cols, DT, DX, dtype=dtype, device=device, boundary=boundary)
    eps, mu = make_materials(fd, materials, rows, cols, dtype=dtype)
    if eps is None:
        eng.set_materials()
    else:
        eng.set_materials(eps, mu, allow_uniform=(materials != "array"))
    del eps, mu
    if boundary == "pml":
        eng.set_pml()
    if not autotune:
        eng.set_option(autotune=False)
    if ENGINE_OPTS:
        eng.set_option(**ENGINE_OPTS)
    for nt, shp in (shapes or {}).items():       # {pass length: (band rows, waves, edge rows, waves side by side, xcd map)}
        if shp and shp[0]:
            eng.set_shape(shp, int(nt) if int(nt) != eng.cycle_steps else 0)
    return eng


def time_single(fd, rows, cols, steps, warmup, materials, device, boundary="mur", shapes=None, autotune=True,
                dtype=np.float32, repeats=11, gpu=None, clock_groups=0):
    """One whole-grid engine on one GPU: `repeats` timed runs of `steps` steps, then the full-length pass kernel
    by itself.  Returns a dict of raw measurements."""
    import torch
    eng = make_engine(fd, rows, cols, materials, device, boundary, shapes, autotune, dtype)
    sr, sc = rows // 2, cols // 2
    eng.prepare(steps, sr, sc)  # launch-shape tuner (trial launches, state untouched): part of set-up
    eng.run(warmup, sr, sc, amplitudes(fd, 0, warmup)).sync()
    cyc = eng.cycle_steps
    copy = [eng.measure_copy(4)]        # what a plain copy of the same arrays reaches on this device right now
    walls, events, first = [], [], warmup
    if gpu is not None:
        gpu.sample()
    warm_ms = clock_warmup(eng, cyc)
    l0 = eng.info(16), eng.info(17)
    for _ in range(repeats):
        amps = amplitudes(fd, first, steps)
        first += steps
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.timer_start()
        eng.run(steps, sr, sc, amps)
        ev_ms = eng.timer_stop()
        eng.sync()
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        events.append(ev_ms)
    res = dict(walls=walls, events_ms=events, pass_launches=(eng.info(16) - l0[0]) // repeats,
               step_launches=(eng.info(17) - l0[1]) // repeats, bpc=eng.bytes_per_cell_step, launch_steps=cyc,
               run_shape=list(eng.last_shape), run_last_nt=eng.last_pass_steps, clock_warmup_ms=warm_ms)
    # duration of the full-length pass kernel by itself: single launches, each between its own pair of HIP
    # events on the engine's stream (what rocprofv3's kernel trace reports)
    if cyc:
        eng.run(cyc).sync()
        res["full_shape"] = list(eng.last_shape)
        one = np.sort(eng.time_launches(48, cyc))
        res["launch_ms"] = float(np.mean(one[4:-4]))       # trimmed mean of back-to-back launches
        res["launch_ms_minmax"] = [float(one[0]), float(one[-1])]
    copy.append(eng.measure_copy(4))
    res["copy_gbps"] = [round(v, 1) for v in copy]
    if gpu is not None:
        res["gpu_during"] = gpu.summary()
    if cyc and clock_groups:
        # launch time against the shader clock the chip really holds meanwhile (in-kernel stamps of a probe that
        # runs beside the launches): groups of 8 full-length launches, first on the run's own fields (zero but for
        # the pulse), then on pseudo-random fields everywhere -- the chip's power management holds a lower clock
        # the more the data toggles (MI355X_MICROARCH.md, DVFS give-back)
        def groups(n):
            out = []
            for _ in range(n):
                eng.clock_probe_start(int(8 * res["launch_ms"] * 1000 * 0.9))
                ms = float(np.mean(eng.time_launches(8, cyc)))
                mhz = [v for v in eng.clock_probe_read() if v > 0]
                out.append([round(float(np.mean(mhz)), 1) if mhz else None, round(ms, 5)])
            return out
        res["clock_vs_launch"] = {"fields_of_the_run": groups(clock_groups)}
        eng.upload(hash_rows(0, rows, cols, 1, 1.0, dtype), hash_rows(0, rows, cols, 2, 1e-3, dtype)[:, :cols - 1],
                   hash_rows(0, rows - 1, cols, 3, 1e-3, dtype))
        eng.run(2 * cyc).sync()
        res["clock_vs_launch"]["random_fields"] = groups(clock_groups)
    Ez, _, _ = eng.download()
    assert np.isfinite(Ez).all() and np.abs(Ez).max() > 0, "benchmark produced an empty field"
    del Ez
    eng.close()
    return res


# ---- HBM traffic of the pass kernels, measured by this run ---------------------------------------------

def _shape_args(shapes):
    out = []
    for nt, shp in (shapes or {}).items():
        if shp and shp[0]:
            out += ["--shape", ":".join(str(int(v)) for v in [nt, *shp])]
    return out


def pmc_child(args):
    """`bench.py --pmc-child`: the K-step plan of the parent (warm runs, then 6 measured ones) and 8 full-length
    passes, the sections separated by a k_reduce dispatch as a marker in the kernel trace.  Run plain (to find the
    launch shapes on the quiet GPU) and under `rocprofv3 --pmc ...` by measure_traffic()."""
    import fdtd2d_amd as fd
    shapes = {}
    for s in args.shape or []:
        nt, *rest = (int(v) for v in s.split(":"))
        shapes[nt] = tuple(rest)
    eng = make_engine(fd, args.grid, args.cols, args.materials, 0, args.boundary, shapes, dtype=NP_DTYPE[args.dtype])
    K, cyc = args.child_steps, eng.cycle_steps
    sr, sc = args.grid // 2, args.cols // 2
    amps = amplitudes(fd, 0, max(K, 8 * max(cyc, 1)))
    eng.prepare(K, sr, sc)
    for _ in range(3):
        eng.run(K, sr, sc, amps)
    eng.sync()
    found = {eng.last_pass_steps: list(eng.last_shape)}
    eng.reduce("Ez")                                   # marker
    l0 = eng.info(16)
    for _ in range(6):
        eng.run(K, sr, sc, amps)
    eng.sync()
    passes = (eng.info(16) - l0) // 6
    eng.reduce("Ez")                                   # marker
    if cyc:
        eng.run(8 * cyc, sr, sc, amps).sync()
        found[cyc] = list(eng.last_shape)
    eng.reduce("Ez")                                   # marker
    print(json.dumps({"pmc_child": True, "shapes": found, "cycle": cyc, "passes_per_run": passes}), flush=True)
    eng.close()


def _child_cmd(child_args):
    return [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", *child_args]


def _child_info(stdout):
    for line in stdout.splitlines():
        if line.startswith("{") and "pmc_child" in line:
            info = json.loads(line)
            info["shapes"] = {int(k): v for k, v in info["shapes"].items()}
            return info
    return None


def _pmc_run(counters, child_args, timeout=300):
    """One rocprofv3 PMC pass over a child -> ([(kernel name, {counter: value}) in dispatch order], child info)."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        raise RuntimeError("rocprofv3 not on PATH")
    out = tempfile.mkdtemp(prefix="fdtd2d_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", out, "--", *_child_cmd(child_args)]
        p = subprocess.run(cmd, cwd=out, capture_output=True, text=True, timeout=timeout)
        info = _child_info(p.stdout)
        disp = {}
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                d = disp.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], {}])
                d[1][r["Counter_Name"]] = d[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if info is None or not disp:
            raise RuntimeError(f"rocprofv3 child produced no counters (exit code {p.returncode}): "
                               f"{(p.stderr or p.stdout)[-300:]}")
        return [tuple(disp[k]) for k in sorted(disp)], info
    finally:
        shutil.rmtree(out, ignore_errors=True)


def _sections(dispatches):
    """Split the dispatch list at the k_reduce markers -> the sections between them."""
    secs, cur = [], []
    for name, ctr in dispatches:
        if "k_reduce" in name:
            if cur:
                secs.append(cur)
            cur = []
        else:
            cur.append((name, ctr))
    return secs     # (what follows the last marker is dropped)


def _short(name):
    return name.split("(")[0].replace("void ", "").strip()


def _is_pass(name):
    return any(t in name for t in ("k_bulk", "k_pass", "k_zone", "k_update", "k_frame", "k_add_point", "k_probe"))


def measure_traffic(rows, cols, materials, boundary, dtype, steps):
    """A plain child on the quiet GPU fixes the launch shapes; two PMC passes (FETCH_SIZE does not fit beside
    WRITE_SIZE: MI355X_MICROARCH.md, PMC slots) and the caller then run exactly those shapes."""
    base = ["--grid", str(rows), "--cols", str(cols), "--materials", materials, "--boundary", boundary,
            "--dtype", dtype, "--child-steps", str(steps)]
    for k, v in ENGINE_OPTS.items():
        base += ["--opt", f"{k}={v}"]
    p = subprocess.run(_child_cmd(base), capture_output=True, text=True, timeout=300)
    tuned = _child_info(p.stdout)
    if tuned is None:
        raise RuntimeError(f"tuning child failed (exit code {p.returncode}): {(p.stderr or p.stdout)[-300:]}")
    shapes = tuned["shapes"]
    base += _shape_args(shapes)
    d1, info = _pmc_run(["FETCH_SIZE", "SQ_INSTS_VALU"], base)
    d2, _ = _pmc_run(["WRITE_SIZE"], base)
    s1, s2 = _sections(d1), _sections(d2)
    if len(s1) < 2 or len(s2) < 2:
        raise RuntimeError("marker dispatches not found in the counter trace")

    def block(sec_f, sec_w, per):
        kern = {}
        for name, _ in sec_f:
            if _is_pass(name):
                kern[_short(name)] = kern.get(_short(name), 0) + 1
        tot = lambda sec, c: sum(ctr.get(c, 0.0) for name, ctr in sec if _is_pass(name))
        rd, wr = 2.0 * tot(sec_f, "FETCH_SIZE") * 1024 / per, tot(sec_w, "WRITE_SIZE") * 1024 / per
        return {"bytes": int(rd + wr), "read": int(rd), "write": int(wr), "valu_insts": tot(sec_f, "SQ_INSTS_VALU") / per,
                "kernels": [{"name": k, "dispatches": round(v / per, 2)} for k, v in kern.items()]}
    out = {"run": block(s1[-2], s2[-2], 6.0), "passes_per_run": info["passes_per_run"], "shapes": shapes,
           "cycle": info["cycle"],
           "source": "rocprofv3 --pmc FETCH_SIZE (x2) / WRITE_SIZE, separate passes, collected by this run"}
    if info["cycle"]:
        out["full"] = block(s1[-1], s2[-1], 8.0)
    return out


def roofline_block(cells, steps, r, traffic):
    """Roofline of the kernels behind `value` + the full-length pass kernel by itself (module docstring)."""
    ev = float(np.median(r["events_ms"]))
    passes = r["pass_launches"]
    bpc = r["bpc"]
    alg_bytes = cells * steps * bpc
    out = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
           "traffic_source": None, "overfetch": None, "valu_frac": None,
           "kernel": None, "kernels": None, "passes_per_run": passes, "steps_per_run": steps,
           "run_event_ms": {"median": round(ev, 5), "min": round(min(r["events_ms"]), 5), "max": round(max(r["events_ms"]), 5)},
           # per launch: a run of K steps is `passes` passes, each one launch (plus, for 20-step passes, its zone kernel beside it)
           "avg_launch_ms": round(ev / max(1, passes), 5), "steps_per_launch": round(steps / max(1, passes), 2),
           "launch_shape": dict(zip(SHAPE_KEYS, r["run_shape"]), pass_steps=r["run_last_nt"]),
           "algorithmic": {"bytes_per_cell_step": bpc, "bytes_per_run": int(alg_bytes),
                           "rate_GBps": round(alg_bytes / (ev * 1e-3) / 1e9, 1),
                           "x_peak": round(alg_bytes / (ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)}}
    once = cells * bpc      # one read + one write of every field, one read of every coefficient array
    if isinstance(traffic, dict):
        t = traffic["run"]
        real = t["bytes"] / (ev * 1e-3) / 1e9
        names = [k["name"] for k in t["kernels"]]
        npass = max(1, traffic["passes_per_run"])
        out.update(achieved=round(real, 1), frac=round(real / HBM_PEAK_GBS, 4), traffic=int(t["bytes"] / npass),
                   traffic_per_run=t["bytes"], traffic_source=traffic["source"],
                   traffic_read_write=[int(t["read"] / npass), int(t["write"] / npass)],
                   overfetch=round(t["bytes"] / (once * max(1, traffic["passes_per_run"])), 3),
                   kernel=" + ".join(names), kernels=t["kernels"])
        if t["valu_insts"]:
            out["valu_frac"] = round(t["valu_insts"] * VALU_ISSUE_CYCLES / (SIMDS * ev * 1e-3 * PEAK_GHZ * 1e9), 4)
            out["valu_insts_per_run"] = int(t["valu_insts"])
    elif isinstance(traffic, str):
        out["traffic_note"] = traffic
    if "launch_ms" in r:
        ms, spl = r["launch_ms"], r["launch_steps"]
        ss = {"steps_per_launch": spl, "avg_launch_ms": round(ms, 5), "launch_ms_min_max": [round(v, 5) for v in r["launch_ms_minmax"]],
              "value": round(cells * spl / (ms * 1e-3) / 1e6, 1),
              "launch_shape": dict(zip(SHAPE_KEYS, r["full_shape"])),
              "algorithmic_x_peak": round(cells * spl * bpc / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)}
        if isinstance(traffic, dict) and "full" in traffic and traffic["cycle"] == spl:
            t = traffic["full"]
            real = t["bytes"] / (ms * 1e-3) / 1e9
            ss.update(kernel=" + ".join(k["name"] for k in t["kernels"]), traffic=t["bytes"],
                      traffic_read_write=[t["read"], t["write"]], achieved=round(real, 1),
                      frac=round(real / HBM_PEAK_GBS, 4), overfetch=round(t["bytes"] / once, 3))
            if t["valu_insts"]:
                ss["valu_frac"] = round(t["valu_insts"] * VALU_ISSUE_CYCLES / (SIMDS * ms * 1e-3 * PEAK_GHZ * 1e9), 4)
        out["steady_state"] = ss
    return out


def single_record(fd, rows, cols, steps, warmup, materials, boundary, device, pmc, note="", dtype="f32",
                  autotune=True, repeats=11, gpu=None, traffic=None, shapes=None):
    """Timing of one whole-grid configuration + its roofline block.  traffic: the live PMC result (dict) or an
    error string if the live measurement failed, or None (pmc off).  shapes: launch shapes given on the command
    line (else those the PMC children ran)."""
    if shapes is None and isinstance(traffic, dict):
        shapes = traffic["shapes"]
    r = time_single(fd, rows, cols, steps, warmup, materials, device, boundary, shapes, autotune, NP_DTYPE[dtype],
                    repeats, gpu, clock_groups=6 if gpu is not None else 0)
    cells = rows * cols
    wall = float(np.median(r["walls"]))
    rec = {"value": round(cells * steps / wall / 1e6, 1), "unit": "Mcell-steps/s", "steps": steps,
           "warmup": warmup, "ms_per_step": round(wall * 1e3 / steps, 5), "repeats": repeats,
           "value_min_max": [round(cells * steps / max(r["walls"]) / 1e6, 1), round(cells * steps / min(r["walls"]) / 1e6, 1)],
           "wall_ms_all": [round(w * 1e3, 4) for w in r["walls"]], "dtype": dtype,
           "clock_warmup": {"ms": r["clock_warmup_ms"],
                            "how": "untimed full-length passes right before the timed repetitions, after the W warm-up steps and "
                                   "the set-up around them: the chip needs 10-20 ms of work to hold its clock"},
           "config": {"workload": f"{rows}x{cols} {'fp32' if dtype == 'f32' else 'fp64'} TE-mode, {materials} eps/mu, "
                                  + ("Mur-5" if boundary == "mur" else "split-field PML (40 cells)")
                                  + " boundary, ricker point source at the centre" + note,
                      "grid": [rows, cols], "materials": materials, "boundary": boundary},
           "roofline": roofline_block(cells, steps, r, traffic)}
    rl = rec["roofline"]
    rl["copy_kernel_GBps_before_after"] = r["copy_gbps"]
    # the same rates against what a plain copy of the same arrays reaches on this box (the practical ceiling)
    copy = max(r["copy_gbps"]) if r["copy_gbps"] else 0
    if copy > 0:
        if rl.get("achieved"):
            rl["frac_of_copy_rate"] = round(rl["achieved"] / copy, 4)
        if rl.get("steady_state", {}).get("achieved"):
            rl["steady_state"]["frac_of_copy_rate"] = round(rl["steady_state"]["achieved"] / copy, 4)
    if "gpu_during" in r:
        rec["gpu_during"] = r["gpu_during"]
    if "clock_vs_launch" in r:
        c = r["clock_vs_launch"]
        for k in list(c):
            pts = [(m, t) for m, t in c[k] if m]
            c[k] = {"mhz_ms_per_8_launch_group": c[k],
                    "ms_x_ghz": [round(m * t / 1e3, 4) for m, t in pts],
                    "value": round(cells * r["launch_steps"] / (float(np.mean([t for _, t in c[k]])) * 1e-3) / 1e6, 1)}
        rec["roofline"]["steady_state"]["clock_vs_launch"] = c
    return rec


# ---- N > 1: checking the rows next to the cuts ----------------------------------------------------------

def hash_rows(r0, r1, cols, salt, scale, dtype=np.float32):
    """A deterministic pseudo-random field as a pure function of the GLOBAL cell index: every rank (and the
    single-engine check) can produce any rows of it independently."""
    i = np.arange(r0, r1, dtype=np.uint32)[:, None]
    j = np.arange(cols, dtype=np.uint32)[None, :]
    h = i * np.uint32(2654435761) ^ (j * np.uint32(40503) + np.uint32(salt))
    h ^= h >> np.uint32(15)
    h *= np.uint32(2246822519)
    h ^= h >> np.uint32(13)
    return ((h & np.uint32(0xFFFF)).astype(dtype) / dtype(32768) - dtype(1)) * dtype(scale)


def check_cut_bands(fd, runner, rows, cols, materials, boundary, device, owned, steps, src, amps, init, band=24):
    """This rank's rows within `band` of each of its cuts, as the slab run left them (`owned` = downloaded Ez, Hx,
    Hy of the owned rows), against a single-engine run of the sub-grid of rows around the cut: same materials,
    same initial state (`init`: "zero" or "hash"), same source, same steps.  The sub-grid's own top / bottom
    boundary is `steps` + margin rows away from the band -- outside its domain of dependence -- so inside the band
    the two must agree bit for bit iff the halos arrived in the right rows at the right times.
    Returns (identical, rows compared)."""
    L = 40 if boundary == "pml" else 0
    W = steps + band + L + 8
    r0, r1 = runner.r0, runner.r1
    ok, compared = True, 0
    for cut in ([r0] if runner.up is not None else []) + ([r1] if runner.down is not None else []):
        a, b = max(0, cut - W), min(rows, cut + W)
        eps, mu = make_materials(fd, materials, rows, cols, a, b)
        eng = fd.Engine(b - a, cols, DT, DX, dtype=np.float32, device=device, boundary=boundary)
        if eps is None:
            eng.set_materials()
        else:
            eng.set_materials(eps, mu, allow_uniform=(materials != "array"))
        if boundary == "pml":      # the sub-grid is interior in the row direction: column layers only
            courant = (1 / np.sqrt(fd.EPS0 * fd.MU0) * DT) / DX
            P = dict(fd.pml_profiles(b - a, cols, courant, 40, 3, 1e-6, np.float32))
            for k in ("ahr", "bhr", "aer", "ber"):
                if a > 0 and b < rows:
                    P[k] = np.ones(b - a, np.float32)
            eng.set_pml(profiles=P)
        if init == "hash":
            eng.upload(hash_rows(a, b, cols, 1, 1.0), hash_rows(a, b, cols, 2, 1e-3)[:, :cols - 1],
                       hash_rows(a, min(b, rows - 1), cols, 3, 1e-3)[:b - a - 1])
        in_sub = a <= src[0] < b
        eng.run(steps, src[0] - a if in_sub else 0, src[1], amps if in_sub else None)
        sub = eng.download()
        eng.close()
        lo, hi = max(r0, cut - band), min(r1, cut + band)          # this rank's side of the band
        for mine, ref, name in zip(owned, sub, ("Ez", "Hx", "Hy")):
            hi_f = min(hi, rows - 1) if name == "Hy" else hi
            x, y = mine[lo - r0:hi_f - r0], ref[lo - a:hi_f - a]
            compared += x.shape[0]
            if not np.array_equal(x, y):
                ok = False
                bad = np.argwhere(x != y)
                print(f"[rank {runner.rank}] cut {cut}: {name} differs from the single-engine sub-grid run at "
                      f"{len(bad)} cells, first (row {lo + bad[0][0]}, col {bad[0][1]})", file=sys.stderr, flush=True)
    return ok, compared


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--repeats", type=int, default=11, help="N=1: timed repetitions of the K steps (median reported)")
    ap.add_argument("--grid", type=int, default=0, help="rows (default 16384 on one GPU, 4096 per GPU otherwise)")
    ap.add_argument("--cols", type=int, default=0, help="columns (default = rows on one GPU, 4096*gpus otherwise)")
    ap.add_argument("--materials", choices=["uniform", "array", "ring"], default="uniform")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--boundary", choices=["mur", "pml"], default=None,
                    help="mur = the reference's 5-px Mur frame; pml = the build-defined split-field PML "
                         "(default: mur, and pml for --gpus 8 = BASELINE configs[4])")
    ap.add_argument("--pmc", choices=["live", "off"], default="live",
                    help="roofline.traffic: measured by this run through rocprofv3 child processes, or omitted")
    ap.add_argument("--exchange", choices=["overlapped", "plain"], default="overlapped")
    ap.add_argument("--loop", choices=["auto", "c", "python"], default="auto",
                    help="N > 1: the run loop in C with the library's RCCL transport (auto on nccl) or the "
                         "Python-sequenced cycle over torch.distributed")
    ap.add_argument("--no-secondary", action="store_true", help="N=1: skip the secondary records")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="N>1: skip the cut-band check (timing experiments only)")
    ap.add_argument("--no-autotune", action="store_true",
                    help="fixed launch-shape rules (for profiler runs: the tuner's trial launches would be "
                         "averaged into the per-kernel statistics); combine with --shape")
    ap.add_argument("--shape", action="append",
                    help="pass length:band rows:waves:edge band rows[:waves side by side:xcd map:filler rows:fillers per strip] (repeatable)")
    ap.add_argument("--opt", action="append", default=[], help="Engine.set_option knob, name=int (repeatable; experiments)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--child-steps", type=int, default=16, help=argparse.SUPPRESS)
    args = ap.parse_args()

    for o in args.opt:
        ENGINE_OPTS[o.split("=")[0]] = int(o.split("=")[1])
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
        default_cfg = (rows, cols, args.materials, boundary, args.dtype) == (16384, 16384, "uniform", "mur", "f32")
        recs = [(rows, cols, args.steps, args.warmup, args.materials, boundary, args.dtype,
                 " (the grid of BASELINE's 1-GPU target)" if default_cfg else "")]
        if default_cfg and not args.no_secondary:
            recs += [(4096, 4096, 400, 40, "uniform", "mur", "f32", " (BASELINE configs[1])"),
                     (8192, 8192, 160, 32, "ring", "mur", "f32", " (BASELINE configs[2] geometry)"),
                     (16384, 16384, 160, 32, "array", "mur", "f32", " (eps and mu as arrays: 32 B per cell-step)"),
                     (4096, 32768, 160, 32, "uniform", "pml", "f32", " (one rank's slab of BASELINE configs[4], whole-grid PML)"),
                     (4096, 4096, 160, 32, "uniform", "mur", "f64", " (the reference's own arithmetic type)"),
                     (16384, 16384, 48, 16, "uniform", "mur", "f64", " (the reference's own arithmetic type)")]
        # HBM traffic first: the rocprofv3 child processes must start before THIS process has
        # initialised the GPU (an exec from a GPU-initialised process is refused on this pool)
        traffic = []
        for (r_, c_, st, wu, mat, bnd, dt_, note) in recs:
            t = None
            if args.pmc == "live":
                try:
                    t = measure_traffic(r_, c_, mat, bnd, dt_, st)
                except Exception as exc:
                    t = f"live PMC failed ({exc})"
            traffic.append(t)
        # the tolerance build (libfdtd2d_fused.so: multiply-add pairs as FMA) on the headline workload: a child process
        # with FDTD2D_ARITHMETIC=fused -- the library is chosen at import --, also started before this process
        # touches the GPU
        fused_rec = None
        if default_cfg and not args.no_secondary and os.environ.get("FDTD2D_ARITHMETIC", "exact") == "exact" and \
                os.path.exists(os.path.join(ROOT, "fdtd-2d_amd", "libfdtd2d_fused.so")):
            try:
                cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(args.steps), "--warmup", str(args.warmup),
                       "--repeats", str(args.repeats), "--pmc", args.pmc, "--no-secondary", "--no-cpu-baseline"]
                p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, FDTD2D_ARITHMETIC="fused"))
                d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
                fused_rec = {k: d[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "repeats", "value_min_max", "roofline")}
                fused_rec["dtype"] = "f32"
                fused_rec["arithmetic"] = "fused multiply-add (libfdtd2d_fused.so, FDTD2D_ARITHMETIC=fused): within the stated " \
                                          "float32 tolerance of the reference, not value-identical (tests/test_fused_build.py)"
                fused_rec["config"] = dict(d["config"], workload=d["config"]["workload"] + " -- tolerance build")
            except Exception as exc:
                fused_rec = {"arithmetic": "fused", "error": f"{type(exc).__name__}: {exc}"}
        gpu = GpuState(local)
        before = gpu.read()
        import torch
        import fdtd2d_amd as fd
        torch.cuda.set_device(local)
        forced = None
        if args.shape:
            forced = {int(s.split(":")[0]): tuple(int(v) for v in s.split(":")[1:]) for s in args.shape}
        out = []
        for k, ((r_, c_, st, wu, mat, bnd, dt_, note), t) in enumerate(zip(recs, traffic)):
            out.append(single_record(fd, r_, c_, st, wu, mat, bnd, local, args.pmc, note, dt_, not args.no_autotune,
                                     args.repeats, gpu if k == 0 else None, t, forced if k == 0 else None))
        head = out[0]
        res = {"metric": "Mcell-steps/s", "value": head["value"], "unit": "Mcell-steps/s",
               "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
               "data": "synthetic", "config": head["config"], "repeats": head["repeats"],
               "value_min_max": head["value_min_max"], "wall_ms_all": head["wall_ms_all"],
               "clock_warmup": head["clock_warmup"], "roofline": head["roofline"],
               "gpu_state": {"source": "amdsmi gpu_metrics", "partition": getattr(gpu, "partition", None),
                             "device": getattr(gpu, "device", None),
                             "before": before, "during_timed_region": head.get("gpu_during"), "after": gpu.read()}}
        if len(out) > 1 or fused_rec:
            for o in out[1:]:
                o.pop("wall_ms_all", None)
            res["secondary"] = out[1:] + ([fused_rec] if fused_rec else [])
        res["arithmetic"] = os.environ.get("FDTD2D_ARITHMETIC", "exact") + \
            (" (one rounding per operation: value-identical to the reference)" if os.environ.get("FDTD2D_ARITHMETIC", "exact") == "exact" else "")
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
    gpu = GpuState(local) if rank == 0 else None
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
        cycle = runner.cycle
        runner.prepare(args.steps)      # kernels of the last, shorter cycle: part of set-up
        # the last warm-up steps are a run of exactly the timed run's length: the same kernels, cycles and messages once
        # before they are timed (the N = 1 line's first repetition is the slow one for the same reason; there the median of 11
        # takes care of it, here the timed region is a single run)
        runner.run(args.steps, sr, sc, amplitudes(fd, wu, args.steps))
        wu += args.steps
        amps = amplitudes(fd, wu, args.steps)
        if gpu is not None:
            gpu.sample()
        torch.cuda.synchronize()
        dist.barrier()
        # every rank keeps its GPU busy for ~30 ms (copy kernel: the slab's state and halos stay as they are) so that the
        # timed cycles run at the clock a long run holds, like the N = 1 line's repetitions
        warm_ms = clock_warmup(runner.engine)
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
        gpu_during = gpu.summary() if gpu is not None else None
        slab = runner.engine.nrows
        exchange_mode = "overlapped" if runner.overlap and rows >= 2 * (2 * cycle + 6) else "plain"
        loop_used = runner.loop + (" (library RCCL transport)" if runner.loop == "c" and backend == "nccl" else "")
        ranks_seen = runner.engine.slab_ranks() if runner.loop == "c" and hasattr(runner.engine, "slab_ranks") else None
        owned = runner.download()
        finite = all(bool(np.isfinite(a).all()) for a in owned)
        # (1) the timed run itself: the source sits on the middle cut, so the rows next to that cut hold the
        #     pulse -- compared with a single-engine run of the rows around it from the same zero state
        # (2) a second, untimed run through the same runner from a pseudo-random state: EVERY cut sees data
        verify = {"cut_bands_identical": None}
        if not args.no_verify:
            total = wu + args.steps
            ok1, n1 = (True, 0)
            if total <= 1024:
                ok1, n1 = check_cut_bands(fd, runner, rows, cols, args.materials, boundary, local, owned, total,
                                          (sr, sc), amplitudes(fd, 0, total), "zero")
            del owned
            vsteps = 2 * max(cycle, 8) + 8
            r0, r1 = runner.r0, runner.r1
            runner.upload(hash_rows(r0, r1, cols, 1, 1.0), hash_rows(r0, r1, cols, 2, 1e-3)[:, :cols - 1],
                          hash_rows(r0, min(r1, rows - 1), cols, 3, 1e-3))
            vamps = amplitudes(fd, 0, vsteps)
            runner.run(vsteps, sr, sc, vamps)
            ok2, n2 = check_cut_bands(fd, runner, rows, cols, args.materials, boundary, local, runner.download(),
                                      vsteps, (sr, sc), vamps, "hash")
            flag = torch.tensor([1.0 if (ok1 and ok2) else 0.0, float(n1 + n2)],
                                device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(flag[:1], op=dist.ReduceOp.MIN)
            dist.all_reduce(flag[1:], op=dist.ReduceOp.SUM)
            verify = {"cut_bands_identical": bool(flag[0].item() == 1.0), "rows_compared": int(flag[1].item()),
                      "how": f"rows within 24 of every cut vs a single-engine run of the sub-grid around the cut: the "
                             f"timed run's own state at the middle cut ({total} steps from zero fields"
                             + ("" if total <= 1024 else "; skipped, > 1024 steps") + f") and a {vsteps}-step run of "
                             "the same runner from a pseudo-random state at every cut"}
        runner.close()
    except Exception:
        print(f"[rank {rank}] slab run failed:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
        os._exit(3)
    bad = verify["cut_bands_identical"] is False or not finite
    if rank == 0:
        single = time_single(fd, slab, cols, min(args.steps, 96), 16, args.materials, local, boundary, repeats=3)
        single_v = slab * cols * min(args.steps, 96) / float(np.median(single["walls"])) / 1e6
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
                       "per_gpu_slab": [slab, cols], "fields_finite": bool(finite),
                       "exchange": exchange_mode, "cycle_steps": cycle, "loop": loop_used,
                       "rccl_rank_of_ranks": list(ranks_seen) if ranks_seen else None,
                       "torch_world_size": dist.get_world_size(),
                       "host_us_per_cycle": round(host_s * 1e6 / ncyc, 1)},
            "clock_warmup": {"ms": warm_ms, "how": "untimed copy-kernel launches on every rank between two barriers right "
                                                   "before the timed region (state untouched)"},
            "verification": verify,
            "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": None, "traffic": None,
                         "kernel": "pass kernels of all ranks (whole job); per-kernel figures: the N=1 line",
                         "algorithmic": {"bytes_per_cell_step": bpc, "rate_GBps": round(alg, 1),
                                         "x_peak": round(alg / (HBM_PEAK_GBS * world), 3)}},
            # (one rank's slab run alone on rank 0's GPU, untimed part of this job: a yardstick for the reader; the scaling
            # efficiency itself is the driver's to compute from its own per-N runs)
            "single_gpu_same_slab": {"value": round(single_v, 1), "unit": "Mcell-steps/s"},
            "gpu_state": {"source": "amdsmi gpu_metrics, rank 0", "during_timed_region": gpu_during},
        }
        print(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()
    if bad:
        sys.exit(4)


if __name__ == "__main__":
    main()
