"""N row-slab HIP engines in ONE process, one thread per rank -- TEST INFRASTRUCTURE.

The GPU box allows at most 6 processes on its card, so BASELINE configs[4]'s 8-way decomposition
cannot be rehearsed with one process per rank there.  Here every rank is a thread that drives its
own handle through the library's run loop in C (fdtd2d_run_slab); the attached transport is a
callback that meets the other ranks at a barrier and copies the neighbours' packed send buffers
device-to-device into its own receive buffers.  What runs is exactly what a multi-GPU job runs per
rank -- slab engines, edge pieces on the edge stream, 3- / 4-field pack and unpack, overlapped
cycles, commit -- minus RCCL.  It also exercises the header's promise that distinct handles may be
driven by distinct threads.
"""
import threading

import numpy as np


def run_local_slabs(fd, world, shape, dtype, boundary, st, nsteps, src, *, cycle_opt=None, overlap=True,
                    materials="array", extent=None, pml=None, options=None):
    """Returns (Ez, Hx, Hy) of the whole grid assembled from the ranks' owned rows, plus the cycle used.
    st: dict with full-grid Ez, Hx, Hy, eps, mu, amps (float64; cast here)."""
    import hipmem
    from fdtd2d_amd.slab import plan_slabs, HALO
    rows, cols = shape
    dt_ = np.dtype(dtype)
    halo = HALO if rows // world >= 6 + HALO else 8
    plan = plan_slabs(rows, world, halo)
    engines, bufs = [], []
    corner = (float(st["eps"][0, 0]), float(st["mu"][0, 0]))
    for r0, r1 in plan:
        eng = fd.Engine(rows, cols, st["dt"], st["dx"], dtype=dt_, boundary=boundary, slab=(r0, r1 - r0, halo))
        lo, hi = eng.stored_rows
        if materials == "uniform":
            eng.set_materials(float(st["eps"][0, 0]), float(st["mu"][0, 0]))
        else:
            eng.set_materials(st["eps"][lo:hi].astype(dt_), st["mu"][lo:hi].astype(dt_), corner=corner)
        if boundary == "pml":
            eng.set_pml(**(pml or {}))
        if cycle_opt is not None:
            eng.set_option(max_pass_steps=cycle_opt)
        if options:
            eng.set_option(**options)
        if extent:
            eng.set_source_extent(*extent)
        eng.upload(st["Ez"][r0:r1].astype(dt_), st["Hx"][r0:r1].astype(dt_),
                   st["Hy"][r0:min(r1, rows - 1)].astype(dt_))
        engines.append(eng)
    cycle = min(halo, min(e.cycle_steps for e in engines))
    for e in engines:
        if e.cycle_steps != cycle:
            e.set_option(max_pass_steps=cycle)
    for k, eng in enumerate(engines):
        b = {}
        for side, has in ((0, k > 0), (1, k < world - 1)):
            if has:
                b[side] = (hipmem.DevBuf(eng.halo_bytes), hipmem.DevBuf(eng.halo_bytes))      # (send, recv)
        bufs.append(b)
    hipmem.sync()
    barrier = threading.Barrier(world)
    errors = []

    def transport(k):
        def fn(_st, _rt, _sb, _rb, _nbytes, _stream):
            hipmem.sync()                            # this rank's packs (and everyone else's) are done
            barrier.wait(timeout=120)
            if 0 in bufs[k]:
                bufs[k][0][1].copy_from(bufs[k - 1][1][0])   # my top halo <- upper neighbour's bottom rows
            if 1 in bufs[k]:
                bufs[k][1][1].copy_from(bufs[k + 1][0][0])   # my bottom halo <- lower neighbour's top rows
            hipmem.sync()
            barrier.wait(timeout=120)                # nobody repacks before every copy has been made
            return 0
        return fn

    for k, eng in enumerate(engines):
        eng.slab_attach({s: (b[0].ptr, b[1].ptr) for s, b in bufs[k].items()}, transport(k))
    min_slab = min(b - a for a, b in plan)
    can_overlap = bool(overlap) and min_slab >= 2 * halo + 5 and rows >= 2 * (2 * cycle + 6)

    def work(k):
        try:
            engines[k].run_slab(nsteps, cycle, can_overlap, src[0], src[1], st["amps"])
            engines[k].sync()
        except BaseException as exc:      # a rank that dies must not leave the others at the barrier
            errors.append((k, exc))
            barrier.abort()

    threads = [threading.Thread(target=work, args=(k,)) for k in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        for e in engines:
            e.close()
        raise errors[0][1]
    parts = [e.download() for e in engines]
    launches = [e.info(16) for e in engines]
    for e in engines:
        e.close()
    return tuple(np.concatenate([p[k] for p in parts], axis=0) for k in range(3)), cycle, can_overlap, launches
