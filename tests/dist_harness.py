"""Spawns world_size processes that run a SlabRunner job and checks rank 0's gathered result."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, job, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    cyc = job.get("cycle", 8)      # one value, or one per rank (engines that disagree: see _agree_cycle)
    os.environ["FAKE_ENGINE_CYCLE"] = str(cyc[rank] if isinstance(cyc, (list, tuple)) else cyc)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fdtd2d_amd.slab import SlabRunner
        factory = None
        if job["engine"] == "fake":
            from fake_engine import FakeEngine
            factory = FakeEngine
        rows, cols = job["shape"]
        dtype = np.dtype(job["dtype"])
        runner = SlabRunner(rows, cols, job["dt"], job["dx"], dtype=dtype, device=0,
                            engine_factory=factory, overlap=job.get("overlap", True),
                            boundary=job.get("boundary", "mur"), loop=job.get("loop", "auto"))
        lo, hi = runner.engine.stored_rows
        r0, r1 = runner.engine.owned_rows
        st = np.load(job["state"])
        if job["materials"] == "uniform":
            runner.set_materials(float(st["eps"][0, 0]), float(st["mu"][0, 0]))
        else:
            runner.set_materials(st["eps"][lo:hi].astype(dtype), st["mu"][lo:hi].astype(dtype))
        if job.get("extent"):
            runner.engine.set_source_extent(*job["extent"])
        if job.get("options"):
            runner.set_option(**job["options"])
            assert runner.engine.cycle_steps == job["options"].get("max_pass_steps", 8)
        runner.upload(st["Ez"][r0:r1].astype(dtype), st["Hx"][r0:r1].astype(dtype),
                      st["Hy"][r0:min(r1, rows - 1)].astype(dtype))
        if job.get("probe"):
            runner.set_probe(job["probe"][0], job["probe"][1], sum(job["chunks"]))
        if job.get("expect_cycle") is not None:
            assert runner.cycle == job["expect_cycle"], (runner.cycle, job["expect_cycle"])
        if job.get("expect_overlap") is not None:
            assert runner.overlap == job["expect_overlap"], (rank, runner.overlap)
        done = 0
        for n in job["chunks"]:
            runner.run(n, job["src"][0], job["src"][1], st["amps"][done:done + n])
            done += n
        if job.get("probe"):
            pr = runner.read_probe()
            if pr is not None:
                np.save(os.path.join(outdir, "probe.npy"), pr)
        out = runner.gather(0)
        if rank == 0:
            np.savez(os.path.join(outdir, "result.npz"), Ez=out[0], Hx=out[1], Hy=out[2])
        runner.close()
    finally:
        dist.barrier()
        dist.destroy_process_group()


def run_job(world, job, outdir):
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(worker, args=(world, port, job, outdir), nprocs=world, join=True)
    r = np.load(os.path.join(outdir, "result.npz"))
    return r["Ez"], r["Hx"], r["Hy"]
