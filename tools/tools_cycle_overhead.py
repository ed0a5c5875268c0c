#!/usr/bin/env python3
"""GPU box: cost of the overlapped exchange cycle WITHOUT the transport (mocked), i.e. Python
sequencing + edge launches + pack/unpack, for a middle slab of 4096 x cols."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import fdtd2d_amd as fd
from fdtd2d_amd.slab import SlabRunner

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("gloo", rank=0, world_size=1)
for cols in (8192, 16384, 32768):
    class R(SlabRunner):
        pass
    # build a 3-slab plan but run only the middle rank's engine, transport mocked
    r = SlabRunner.__new__(SlabRunner)
    rows = 3 * 4096
    r.dist, r.torch, r.group, r.rank, r.world = dist, torch, None, 1, 3
    r.rows, r.cols, r.dt, r.dx, r.dtype, r.halo = rows, cols, 5e-14, 1e-4, np.dtype(np.float32), 8
    r.r0, r.r1 = 4096, 8192
    r.engine = fd.Engine(rows, cols, dtype=np.float32, slab=(4096, 4096, 8))
    r.up, r.down, r.backend, r.buf_device, r.boundary = 0, 2, "nccl", "cuda:0", "mur"
    r.stream, r.edge_stream = torch.cuda.Stream(), torch.cuda.Stream()
    r.engine.set_stream(r.stream.cuda_stream)
    n = 3 * 8 * cols
    r._bufs = {s: (torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), None) for s in (0, 1)}
    r.overlap, r._halo_fresh, r.steps_done = True, False, 0
    r._transfer = lambda sides: []           # no transport
    r.engine.set_materials()
    amps = np.zeros(160)
    r.run(16, 6000, 100, amps)
    torch.cuda.synchronize()
    for steps in (160,):
        t0 = time.perf_counter(); r.run(steps, 6000, 100, amps); th = time.perf_counter() - t0
        torch.cuda.synchronize(); tw = time.perf_counter() - t0
        print(f"cols {cols}: overlapped cycle host {th/ (steps/8)*1e6:7.1f} us/cycle, wall {tw/(steps/8)*1e6:7.1f} us/cycle "
              f"-> {4096*cols*steps/tw/1e6:10.0f} Mcell-steps/s", flush=True)
    r.engine.close()
    with fd.Engine(4096, cols, dtype=np.float32) as e:
        e.set_materials(); e.run(16); e.sync()
        t0 = time.perf_counter(); e.run(160); e.sync(); tw = time.perf_counter() - t0
        print(f"cols {cols}: plain engine same slab          wall {tw/20*1e6:7.1f} us/pass  -> {4096*cols*160/tw/1e6:10.0f} Mcell-steps/s", flush=True)
