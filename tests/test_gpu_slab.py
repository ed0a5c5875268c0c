"""GPU box: the real HIP slab engines (two and three processes sharing cuda:0, gloo transport
staged through host memory -- the box has one GPU, RCCL needs one GPU per rank) reproduce the
single-engine result bit for bit: slab engines, halo pack/unpack kernels, validity tracking,
pass geometry with halos, zones on the first/last rank only."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from dist_harness import run_job

DT, DX = 5e-14, 1e-4


@pytest.mark.parametrize("world,shape,dtype,materials,overlap,options", [
    (2, (200, 300), "float32", "array", True, None),
    (2, (200, 300), "float32", "array", True, {"loop": "c"}),               # the run loop in C (fdtd2d_run_slab)
    (3, (300, 1100), "float32", "uniform", True, {"max_pass_steps": 16, "loop": "c"}),
    (3, (180, 520), "float32", "array", False, {"loop": "c"}),
    (4, (160, 520), "float32", "uniform", True, {"max_pass_steps": 16}),    # BASELINE configs[3]: 4 slabs
    (4, (160, 520), "float32", "array", True, {"max_pass_steps": 16, "loop": "c"}),
    (3, (180, 520), "float32", "array", True, None),
    (2, (128, 256), "float64", "array", True, None),
    (2, (160, 700), "float32", "uniform", True, None),
    (3, (180, 520), "float32", "array", False, None),
    (2, (40, 300), "float32", "array", True, None),                         # 8-row halo
    (2, (160, 700), "float32", "uniform", True, {"max_pass_steps": 16}),    # 16 steps per exchange
    (3, (300, 1100), "float32", "uniform", True, {"max_pass_steps": 16}),
    (2, (160, 700), "float32", "uniform", False, {"max_pass_steps": 16}),
    (2, (170, 600), "float32", "array", True, {"max_pass_steps": 16}),
    (2, (170, 600), "float32", "array", True, {"max_pass_steps": 16, "extent": (60, 3)}),
    (3, (180, 520), "float64", "array", True, {"extent": (1, 300)}),
    # slabs large enough for the launch-shape tuner (>= 4 Mi cells per piece): its trial launches run INSIDE the
    # first overlapped cycle, after the rows next to the cuts have been issued
    (2, (2400, 3600), "float32", "uniform", True, {"max_pass_steps": 16, "loop": "c"}),
    (2, (2400, 3600), "float32", "uniform", True, {"max_pass_steps": 16}),
])
def test_gpu_slabs_match_single_engine_and_oracle(tmp_path, world, shape, dtype, materials, overlap,
                                                  options):
    import fdtd2d_amd as fd
    from oracle import fdtd_numpy as onp
    r, c = shape
    options = dict(options or {})
    loop = options.pop("loop", "python")
    n = 45 if options else 29                # exchange cycles 8+8+8+5 (passes 8,8,8,4,1) / 16+16+13
    rng = np.random.default_rng(r + c)
    st = dict(Ez=rng.standard_normal((r, c)), Hx=rng.standard_normal((r, c - 1)) * 1e-3,
              Hy=rng.standard_normal((r - 1, c)) * 1e-3,
              eps=onp.EPS0 * rng.uniform(1, 10, (r, c)), mu=onp.MU0 * np.ones((r, c)),
              amps=rng.standard_normal(n))
    if materials == "uniform":
        st["eps"][:] = 3 * onp.EPS0
    path = os.path.join(str(tmp_path), "state.npz")
    np.savez(path, **st)
    extent = (options or {}).get("extent") or (1, 1)
    src = (r // world, c // 2 - extent[1] // 2)       # on the first cut
    job = dict(engine="hip", shape=shape, dtype=dtype, dt=DT, dx=DX, state=path, src=src,
               chunks=[n], materials=materials, overlap=overlap, loop=loop,
               options={k: v for k, v in (options or {}).items() if k != "extent"} or None,
               extent=(options or {}).get("extent"))
    got = run_job(world, job, str(tmp_path))
    dt_ = np.dtype(dtype)
    with fd.Engine(r, c, DT, DX, dtype=dt_) as eng:
        eng.set_materials(st["eps"].astype(dt_), st["mu"].astype(dt_))
        eng.upload(st["Ez"].astype(dt_), st["Hx"].astype(dt_), st["Hy"].astype(dt_))
        eng.set_source_extent(*extent)
        eng.run(n, src[0], src[1], st["amps"])
        one = eng.download()
    ref = [st[k].astype(dt_) for k in ("Ez", "Hx", "Hy")]
    onp.leapfrog(*ref, st["eps"].astype(dt_), st["mu"].astype(dt_), DT, DX, n, src[0], src[1],
                 amps=st["amps"], extent=extent)
    for a, b, c_, k in zip(got, one, ref, ("Ez", "Hx", "Hy")):
        assert np.array_equal(a, b), f"{k}: slabs differ from the single engine"
        assert np.array_equal(a, c_), f"{k}: slabs differ from the oracle"


@pytest.mark.parametrize("probe_row", [85, 84, 100, 2])
def test_gpu_slabs_probe_next_to_the_cut(tmp_path, probe_row):
    """N4 over two slabs (cut at row 85): the probe on the first owned row of rank 1, on the
    last one of rank 0, further inside, and in rank 0's top zone; overlapped 16-step cycles."""
    from oracle import fdtd_numpy as onp
    r, c, n = 170, 600, 45
    rng = np.random.default_rng(probe_row)
    st = dict(Ez=rng.standard_normal((r, c)), Hx=rng.standard_normal((r, c - 1)) * 1e-3,
              Hy=rng.standard_normal((r - 1, c)) * 1e-3,
              eps=onp.EPS0 * rng.uniform(1, 10, (r, c)), mu=onp.MU0 * np.ones((r, c)),
              amps=rng.standard_normal(n))
    path = os.path.join(str(tmp_path), "state.npz")
    np.savez(path, **st)
    job = dict(engine="hip", shape=(r, c), dtype="float32", dt=DT, dx=DX, state=path, src=(80, 300),
               chunks=[n], materials="array", overlap=True, options={"max_pass_steps": 16},
               probe=(probe_row, 301))
    run_job(2, job, str(tmp_path))
    got = np.load(os.path.join(str(tmp_path), "probe.npy"))
    want = []
    ref = [st[k].astype(np.float32) for k in ("Ez", "Hx", "Hy")]
    onp.leapfrog(*ref, st["eps"].astype(np.float32), st["mu"].astype(np.float32), DT, DX, n, 80, 300,
                 amps=st["amps"], on_step=lambda i, E, *_: want.append(float(E[probe_row, 301])))
    assert np.array_equal(got, np.array(want))


@pytest.mark.parametrize("options,n,loop", [(None, 21, "python"), ({"max_pass_steps": 16}, 45, "python"),
                                            ({"max_pass_steps": 16}, 45, "c")])
def test_gpu_slabs_pml_match_single_engine(tmp_path, options, n, loop):
    """boundary="pml" over 2 slabs (4-field halo messages) equals the single-engine PML run bit for
    bit: 8-step passes + single steps consuming the halo, and overlapped 16-step cycles on the
    level-split PML pair (edge rows first, 4-field pack, interior behind the transfer)."""
    import fdtd2d_amd as fd
    from oracle import fdtd_numpy as onp
    r, c = 220, 260
    rng = np.random.default_rng(5)
    st = dict(Ez=np.zeros((r, c)), Hx=np.zeros((r, c - 1)), Hy=np.zeros((r - 1, c)),
              eps=onp.EPS0 * rng.uniform(1, 3, (r, c)), mu=onp.MU0 * np.ones((r, c)),
              amps=rng.standard_normal(n))
    st["eps"][0, 0] = onp.EPS0
    path = os.path.join(str(tmp_path), "state.npz")
    np.savez(path, **st)
    job = dict(engine="hip", shape=(r, c), dtype="float32", dt=DT, dx=DX, state=path, src=(110, 130),
               chunks=[n], materials="array", boundary="pml", options=options, loop=loop)
    got = run_job(2, job, str(tmp_path))
    with fd.Engine(r, c, DT, DX, dtype=np.float32, boundary="pml") as eng:
        eng.set_materials(st["eps"].astype(np.float32), st["mu"].astype(np.float32))
        eng.set_pml()
        if options:
            eng.set_option(**options)
        eng.run(n, 110, 130, st["amps"])
        one = eng.download()
    assert np.abs(one[0]).max() > 0
    for a, b, k in zip(got, one, ("Ez", "Hx", "Hy")):
        assert np.array_equal(a, b), k


@pytest.mark.parametrize("with_torch", [False, True])
def test_rccl_transport_glue_on_one_gpu(with_torch):
    """The built-in transport of the C loop (dlopen of librccl, ncclCommInitRank, grouped ncclSend / ncclRecv on
    a non-blocking stream, ncclFloat32) cannot run between ranks on a one-GPU box; its glue can: a one-rank
    communicator sending to itself.  with_torch: torch.distributed's "nccl" process group is up first, so the
    library must pick up the RCCL copy torch already loaded (what a real multi-GPU job looks like)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pre = ""
    if with_torch:
        pre = ("import torch, torch.distributed as dist\n"
               "dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29653', rank=0, world_size=1, "
               "device_id=torch.device('cuda', 0))\n"
               "t = torch.ones(4, device='cuda'); dist.all_reduce(t); torch.cuda.synchronize()\n")
    code = (f"import sys; sys.path.insert(0, {root!r})\n" + pre +
            "import fdtd2d_amd as fd\n"
            "fd.Engine.rccl_selftest(0, 3 * 16 * 4096)\n"          # one 16-row halo message of a 4096-column slab
            "libs = [l.split()[-1] for l in open('/proc/self/maps') if 'librccl' in l]\n"
            "print('RCCL_OK', sorted(set(libs)))\n")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RCCL_OK")][-1]
    assert line.count("librccl") == 1, "two RCCL builds in one process: " + line
