"""Row-slab decomposition over several MI355X: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI) as the transport.

Rank r owns global rows [r0, r1) of every field and stores `halo` extra rows on each
interior side.  One exchange sends the `halo` owned rows next to each cut (Ez, Hx, Hy packed
into ONE message per neighbour by a device kernel) and is good for `halo` full time steps:
the engine advances its slab by one temporally blocked pass of `halo` steps, recomputing the
shrinking halo region redundantly, so the ranks talk once per 8 steps instead of twice per
step, with 8x larger messages -- the shape that suits point-to-point xGMI links.  The
arithmetic per cell is unchanged, so the N-GPU result is value-identical to the 1-GPU one.

The path has no collective: neighbours only, send/recv.  Boundary handling: the 5-px Mur
band's left/right part is row-local (every rank); the top/bottom bands and the corners
live entirely inside the first / last rank (the planner keeps every cut at least 6 + halo
rows away from the grid's top and bottom).  The Mur factor and the source amplitude are
host scalars every rank computes identically; the source cell is applied by every rank
whose stored rows contain it (owned or halo), which is what the redundant halo computation
needs.

`SlabRunner` only moves bytes and sequences calls; all field arithmetic happens in the
engine (HIP kernels behind the C ABI).  The engine class is injectable so that the very
same sequencing code is exercised on CPU by tests (gloo, world_size >= 2) with an
oracle-backed stand-in that lives under tests/.
"""
from __future__ import annotations

import numpy as np

HALO = 16  # halo rows stored and exchanged = the longest temporally blocked pass (16 steps for
           # float32 + uniform materials; other configurations run 8 steps per exchange)


def plan_slabs(rows: int, world: int, halo: int = HALO):
    """Contiguous, balanced row ranges [(r0, r1), ...] for `world` ranks.

    Every cut must keep `6 + halo` rows clear of the grid's top and bottom (the horizontal
    Mur band with its inputs, plus the neighbour's halo) and every slab must be at least
    `halo` rows tall (it has to fill a neighbour's halo from owned rows)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    if world == 1:
        return [(0, rows)]
    need = max(6 + halo, halo)
    base, rem = divmod(rows, world)
    if base < need:
        raise ValueError(f"{rows} rows cannot be cut into {world} slabs with halo {halo}: "
                         f"each slab needs at least {need} rows")
    out, r = [], 0
    for k in range(world):
        n = base + (1 if k < rem else 0)
        out.append((r, r + n))
        r += n
    return out


def _torch_dtype(np_dtype):
    import torch
    return {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[np.dtype(np_dtype)]


class SlabRunner:
    """Drives one rank's slab: materials, halo exchange, passes, gather.

    Parameters mirror Engine; `group` is a torch.distributed process group (default: the
    world), `engine_factory(rows, cols, dt, dx, dtype, boundary, device, slab)` builds the
    per-rank engine (default: the HIP Engine)."""

    def __init__(self, rows, cols, dt=5e-14, dx=1e-4, dtype=np.float32, boundary="mur", device=0,
                 halo=None, group=None, engine_factory=None, overlap=True, loop="auto"):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.rows, self.cols, self.dt, self.dx = int(rows), int(cols), float(dt), float(dx)
        self.dtype = np.dtype(dtype)
        if halo is None:     # 16 rows where the slabs are tall enough for them, else 8
            halo = HALO if self.rows // self.world >= 6 + HALO else 8
        self.halo = int(halo) if self.world > 1 else 0
        self.plan = plan_slabs(self.rows, self.world, max(self.halo, 1))
        self.r0, self.r1 = self.plan[self.rank]
        if engine_factory is None:
            from .engine import Engine
            engine_factory = Engine
        slab = None if self.world == 1 else (self.r0, self.r1 - self.r0, self.halo)
        self.engine = engine_factory(self.rows, self.cols, self.dt, self.dx, dtype=self.dtype,
                                     boundary=boundary, device=device, slab=slab)
        self.up = self.rank - 1 if self.rank > 0 else None          # neighbour above (lower rows)
        self.down = self.rank + 1 if self.rank < self.world - 1 else None
        self.backend = dist.get_backend(group)
        self.buf_device = getattr(self.engine, "buffer_device", None) or f"cuda:{device}"
        self._bufs = {}
        # All device work of this rank (engine kernels, pack/unpack, and -- through the
        # stream-ordering rules of torch.distributed -- the RCCL transfers) is ordered on ONE
        # side stream, so the host never has to wait for the GPU inside run().
        self.stream = self.edge_stream = None
        if str(self.buf_device).startswith("cuda"):
            self.stream = torch.cuda.Stream(device=self.buf_device)
            self.edge_stream = torch.cuda.Stream(device=self.buf_device)
            self.engine.set_stream(self.stream.cuda_stream)
        # overlap: compute the rows next to the cuts first (second stream), send them while
        # the interior of the slab is still being computed.  Whether a cycle runs overlapped and
        # how many steps it has are decided from quantities EVERY rank shares (the slab plan, the
        # agreed cycle): ranks that disagreed would post their sends and receives in different
        # orders and deadlock.
        self.boundary = boundary
        self.min_slab = min(b - a for a, b in self.plan)
        # (2 * halo + 5: two edge pieces of `halo` rows, and on the first / last rank an interior piece
        # that holds the whole 5 + cycle rows of the top / bottom zone -- a piece may not cut through it)
        self.overlap = bool(overlap) and self.world > 1 and self.min_slab >= 2 * self.halo + 5 and \
            hasattr(self.engine, "pass_rows")
        self.cycle = None            # steps per exchange, agreed in set_materials()
        self._halo_fresh = False
        if self.world > 1:
            n = self.engine.halo_bytes // self.dtype.itemsize    # 3 fields (4 with the PML's Ezx)
            td = _torch_dtype(self.dtype)
            for side, nb in ((0, self.up), (1, self.down)):
                if nb is None:
                    continue
                send = torch.empty(n, dtype=td, device=self.buf_device)
                recv = torch.empty(n, dtype=td, device=self.buf_device)
                stage = None
                if self.backend == "gloo" and send.is_cuda:   # gloo moves host memory only
                    stage = (torch.empty(n, dtype=td).pin_memory(), torch.empty(n, dtype=td).pin_memory())
                self._bufs[side] = (send, recv, stage)
        self.steps_done = 0
        # loop = "c": the whole run is enqueued by ONE call into the library (fdtd2d_run_slab) -- with
        # the "nccl" backend through the library's own RCCL point-to-point transport (communicator
        # created from an id broadcast here), with gloo through a callback that stages the messages
        # through the host (tests).  loop = "python": the cycle is sequenced below, call by call, with
        # torch.distributed as the transport.  "auto" = "c" on nccl where the engine offers it.
        if loop == "auto":
            loop = "c" if (self.backend == "nccl" and hasattr(self.engine, "run_slab")) else "python"
        self.loop = loop if self.world > 1 else "python"
        if self.loop == "c":
            if self.backend == "nccl":
                uid = torch.zeros(128, dtype=torch.uint8)
                if self.rank == 0:
                    uid = torch.frombuffer(bytearray(self.engine.rccl_unique_id()), dtype=torch.uint8).clone()
                uid = self._host_collective(uid, lambda t: dist.broadcast(t, src=self._global(0), group=self.group))
                self.engine.slab_attach_rccl(bytes(uid.numpy().tobytes()), self.rank, self.world)
            else:
                self.engine.slab_attach({s: (b[0].data_ptr(), b[1].data_ptr()) for s, b in self._bufs.items()},
                                        self._c_transport)

    def _c_transport(self, send_top, recv_top, send_bottom, recv_bottom, nbytes, stream):
        """Transport callback of the C loop for backends that move host memory (gloo): blocking."""
        self.torch.cuda.synchronize()
        for w in self._transfer(self._sides()):
            w.wait()
        return 0

    # -- setup ----------------------------------------------------------------------------
    def set_materials(self, eps=None, mu=None, allow_uniform=True):
        """eps, mu: arrays for THIS rank's stored rows (engine.stored_rows), scalars, or None
        for vacuum.  The [0,0] cell that fixes the Mur factor is broadcast from rank 0."""
        torch, dist = self.torch, self.dist
        from .api import EPS0, MU0
        e_s = EPS0 if eps is None else eps
        m_s = MU0 if mu is None else mu
        scalar = np.isscalar(e_s) and np.isscalar(m_s)
        c = torch.zeros(2, dtype=torch.float64)
        if self.rank == 0:
            c[0] = float(e_s if np.isscalar(e_s) else np.asarray(e_s)[0, 0])
            c[1] = float(m_s if np.isscalar(m_s) else np.asarray(m_s)[0, 0])
        c = self._host_collective(c, lambda t: dist.broadcast(t, src=self._global(0), group=self.group))
        lo = torch.tensor([float(np.min(e_s)), float(np.min(m_s))], dtype=torch.float64)
        lo = self._host_collective(lo, lambda t: dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group))
        courant = (1 / np.sqrt(float(lo[0]) * float(lo[1])) * self.dt) / self.dx
        assert courant <= 1.0, f"Courant stability condition not met: {courant} > 1.0"   # fdtd.py:28
        if scalar:
            self.engine.set_materials(float(e_s), float(m_s))
        else:
            self.engine.set_materials(e_s, m_s, corner=(float(c[0]), float(c[1])),
                                      allow_uniform=allow_uniform)
        if self.boundary == "pml":
            self.engine.set_pml(courant00=(1 / np.sqrt(float(c[0]) * float(c[1])) * self.dt) / self.dx)
        self._agree_cycle()
        return self

    def set_option(self, **kw):
        """Engine.set_option on this rank's engine; the steps per exchange are agreed again."""
        self.engine.set_option(**kw)
        if self.cycle is not None:
            self._agree_cycle()
        return self

    def _agree_cycle(self):
        """Steps per exchange = the longest pass EVERY rank's engine runs (an engine decides 16 vs
        8 from its own slab size, and slabs differ by a row): min over the ranks, handed back to
        the engine so that no rank picks a longer pass."""
        torch, dist = self.torch, self.dist
        local = int(getattr(self.engine, "cycle_steps", self.halo) or 0)
        if self.world > 1:
            t = torch.tensor([float(local)], dtype=torch.float64)
            t = self._host_collective(t, lambda x: dist.all_reduce(x, op=dist.ReduceOp.MIN, group=self.group))
            agreed = int(t[0])
        else:
            agreed = local
        if agreed != local and hasattr(self.engine, "set_option"):
            self.engine.set_option(max_pass_steps=agreed)
        self.cycle = min(self.halo, agreed or self.halo) if self.world > 1 else 0
        return self.cycle

    def _global(self, group_rank):
        if self.group is None:
            return group_rank
        return self.dist.get_global_rank(self.group, group_rank)

    def _host_collective(self, t, fn):
        """Tiny setup-time collectives: NCCL needs device tensors, gloo host ones."""
        if self.backend == "nccl":
            d = t.to(self.buf_device)
            fn(d)
            return d.cpu()
        fn(t)
        return t

    def upload(self, Ez=None, Hx=None, Hy=None):
        """Arrays for the OWNED rows in the reference's shapes."""
        self.engine.upload(Ez, Hx, Hy)
        self._halo_fresh = False
        return self

    # -- the loop ---------------------------------------------------------------------------
    def _transfer(self, sides):
        """Move the packed send buffers of `sides` to the neighbours' recv buffers.
        nccl: enqueued behind the current stream, returns the pending works (wait() on them
        makes the then-current stream wait; the host never blocks).  gloo: blocking, staged
        through host memory when the buffers live on the device (tests only)."""
        dist = self.dist
        ops, staged = [], []
        for side in sides:
            send, recv, stage = self._bufs[side]
            if stage is not None:
                self.torch.cuda.synchronize()
                stage[0].copy_(send)
                s, r = stage
                staged.append((recv, r))
            else:
                s, r = send, recv
            peer = self._global(self.up if side == 0 else self.down)
            ops.append(dist.P2POp(dist.isend, s, peer, self.group))
            ops.append(dist.P2POp(dist.irecv, r, peer, self.group))
        works = dist.batch_isend_irecv(ops) if ops else []
        if self.backend != "nccl":
            for w in works:
                w.wait()
            for recv, r in staged:
                recv.copy_(r)
            if staged:
                self.torch.cuda.synchronize()
            works = []
        return works

    def _sides(self):
        return [s for s, nb in ((0, self.up), (1, self.down)) if nb is not None]

    def exchange(self):
        """Owned edge rows -> neighbours' halos (Ez, Hx, Hy; `halo` rows each way).
        Call under `self._on_stream()`."""
        if self.world == 1:
            return
        eng, sides = self.engine, self._sides()
        for side in sides:
            eng.halo_pack(side, self._bufs[side][0].data_ptr())
        for w in self._transfer(sides):
            w.wait()
        for side in sides:
            eng.halo_unpack(side, self._bufs[side][1].data_ptr())
        self._halo_fresh = True

    def _cycle_overlapped(self, n, src_row, src_col, amps):
        """One pass of n (8 or 16) steps with the exchange for the NEXT pass hidden behind the
        interior: (edge stream) rows next to the cuts -> pack -> send/recv;  (main stream)
        everything else;  then commit and unpack.  Needs fresh halos, leaves fresh halos."""
        torch, eng, sides = self.torch, self.engine, self._sides()
        r0, r1, h = self.r0, self.r1, self.halo
        main, edge = self.stream, self.edge_stream
        if edge is not None:
            edge.wait_stream(main)
            eng.set_stream(edge.cuda_stream)
        ctx = torch.cuda.stream(edge) if edge is not None else self._on_stream()
        with ctx:
            for side in sides:
                lo, hi = (r0, r0 + h) if side == 0 else (r1 - h, r1)
                eng.pass_rows(n, lo, hi, src_row, src_col, amps)
                eng.halo_pack(side, self._bufs[side][0].data_ptr())
            works = self._transfer(sides)
        if edge is not None:
            eng.set_stream(main.cuda_stream)
        lo = r0 + h if self.up is not None else 0
        hi = r1 - h if self.down is not None else self.rows
        eng.pass_rows(n, lo, hi, src_row, src_col, amps)
        for w in works:
            w.wait()                       # main stream waits for the transfers
        if edge is not None:
            main.wait_stream(edge)
        eng.pass_commit()
        for side in sides:
            eng.halo_unpack(side, self._bufs[side][1].data_ptr())
        self._halo_fresh = True

    def _on_stream(self):
        import contextlib
        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def run(self, nsteps, src_row=0, src_col=0, amps=None):
        """nsteps of H -> E -> source (fdtd.py:30-34) on the whole decomposed grid."""
        if amps is not None:
            amps = np.ascontiguousarray(amps, dtype=np.float64)
            if amps.shape[0] < nsteps:
                raise ValueError("amps shorter than nsteps")
        done = 0
        if self.world > 1 and self.cycle is None:
            raise RuntimeError("SlabRunner.set_materials() must be called (by every rank) before run()")
        # steps per exchange: the longest pass every rank's engine runs (16 or 8), see _agree_cycle
        cycle = self.cycle
        # grids below 2*(2*cycle+6) rows have no temporally blocked pass (the engine advances
        # them with its single-step kernels), hence nothing to issue in pieces
        can_overlap = self.overlap and self.rows >= 2 * (2 * cycle + 6)
        if self.loop == "c":
            self.engine.run_slab(nsteps, cycle, can_overlap, src_row, src_col, amps)
            self.steps_done += nsteps
            return self
        with self._on_stream():
            while done < nsteps:
                n = nsteps - done if self.world == 1 else min(cycle, nsteps - done)
                a = None if amps is None else amps[done:done + n]
                if self.world > 1 and not self._halo_fresh:
                    self.exchange()
                if can_overlap and n == cycle and n in (8, 16):
                    self._cycle_overlapped(n, src_row, src_col, a)
                else:
                    self.engine.run(n, src_row, src_col, a)
                    self._halo_fresh = False
                done += n
        self.steps_done += nsteps
        return self

    # -- results ----------------------------------------------------------------------------
    def download(self):
        """This rank's owned rows (Ez, Hx, Hy) as host arrays."""
        return self.engine.download()

    def prepare(self, nsteps):
        """Tune / warm the kernels of the LAST, shorter cycle of run(nsteps) now (the full cycles
        are tuned by the first cycles of a run or of a warm-up).  Leaves the fields untouched."""
        prep = getattr(self.engine, "prepare", None)
        if prep is None:
            return self
        cycle = self.cycle or 0
        tail = nsteps % cycle if cycle else nsteps
        if self.loop == "c":           # (the C loop refreshes the halos itself; warm the tail kernels only)
            return self
        with self._on_stream():
            if self.world > 1 and not self._halo_fresh:
                self.exchange()
            if tail:
                prep(tail)
        return self

    # -- point probe ---------------------------------------------------------
    def set_probe(self, row, col, capacity):
        """Ez[row, col] after every step of the following run() calls; recorded on the rank that
        owns the row (read it there with read_probe)."""
        self._probe_owner = self.r0 <= int(row) < self.r1
        if self.boundary == "pml" and self.world > 1:
            # the PML passes have no probe tile: the owner advances by single steps, and a rank that runs
            # plain cycles while its neighbours run overlapped ones would post its transfers in another
            # order -- every rank calls set_probe, so every rank drops the overlap here
            self.overlap = False
        if self._probe_owner or self.world == 1:
            self.engine.set_probe(row, col, capacity)
        return self

    def read_probe(self, first=0, count=None):
        """The samples on the owning rank, None elsewhere."""
        if not getattr(self, "_probe_owner", self.world == 1):
            return None
        with self._on_stream():
            return self.engine.read_probe(first, count)

    def gather(self, dst=0):
        """Full fields on rank `dst` (None elsewhere).  Test/diagnostic helper."""
        parts = self.engine.download()
        if self.world == 1:
            return parts
        out = [None] * self.world if self.rank == dst else None
        self.dist.gather_object(parts, out, dst=self._global(dst), group=self.group)
        if self.rank != dst:
            return None
        return tuple(np.concatenate([p[k] for p in out], axis=0) for k in range(3))

    def sanity(self) -> bool:
        Ez, Hx, Hy = self.engine.download()
        return bool(np.isfinite(Ez).all() and np.isfinite(Hx).all() and np.isfinite(Hy).all())

    def close(self):
        self.engine.close()


def run_fdtd_distributed(rows, cols, dt, dx, nsteps, eps, mu, source, boundary, dtype,
                         on_frame=None, nframes=200, device=None):
    """run_fdtd() when launched under torch.distributed (one process per GPU).  Every rank
    calls it with the same arguments (eps/mu: None, scalars or FULL arrays, sliced here);
    the full (Ez, Hx, Hy) is returned on rank 0, None elsewhere."""
    import os
    import torch.distributed as dist
    from .api import ricker_amplitude, sinusoidal_amplitude
    if not dist.is_initialized():
        raise RuntimeError("run_fdtd(devices>1) needs torch.distributed: launch one process per "
                           "GPU with `python -m torch.distributed.run --nproc-per-node N ...` and "
                           "call torch.distributed.init_process_group('nccl') first")
    device = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device
    runner = SlabRunner(rows, cols, dt, dx, dtype=dtype, boundary=boundary, device=device)
    lo, hi = runner.engine.stored_rows
    sl = lambda a: a if (a is None or np.isscalar(a)) else np.ascontiguousarray(np.asarray(a)[lo:hi])
    runner.set_materials(sl(eps), sl(mu))
    amps, sr, sc = None, 0, 0
    if source is not None:
        kind, sr, sc, fc, *extent = source
        if extent:
            runner.engine.set_source_extent(*extent[0])
        sr = rows // 2 if sr is None else sr
        sc = cols // 2 if sc is None else sc
        f = {"ricker": ricker_amplitude, "sinusoidal": sinusoidal_amplitude}[kind]
        amps = np.array([f(i * dt, fc) for i in range(nsteps)], dtype=np.float64)
    # the snapshot cadence of fdtd.py:36-38: after every step i with i % (nsteps // nframes) == 0
    # the full Ez is gathered on rank 0 and handed to on_frame there (collective: every rank stops)
    every = max(1, nsteps // nframes) if on_frame is not None else nsteps
    done = 0
    while done < nsteps:
        if on_frame is None:
            n = nsteps - done
        else:
            nxt = ((done + every - 1) // every) * every
            n = min(nsteps - done, nxt - done + 1)
        runner.run(n, sr, sc, None if amps is None else amps[done:done + n])
        done += n
        if on_frame is not None and (done - 1) % every == 0:
            full = runner.gather(0)
            if full is not None:
                on_frame(done - 1, full[0])
    out = runner.gather(0)
    runner.close()
    return out
