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
    # slabs of 35 / 36 rows with forced 16-step cycles: an edge piece would cut through the 21-row top / bottom zone
    # of the first / last rank, so every rank must run plain cycles (round-2 advisor finding)
    (3, (105, 300), "float32", "uniform", True, {"max_pass_steps": 16, "loop": "c", "expect_overlap": False}),
    (3, (108, 300), "float32", "array", True, {"max_pass_steps": 16, "expect_overlap": False}),
    (3, (111, 300), "float32", "uniform", True, {"max_pass_steps": 16, "loop": "c", "expect_overlap": True}),
    # float64 on 16-step cycles (round 3: the reference's dtype has the 16-step kernel too)
    (2, (170, 600), "float64", "array", True, {"max_pass_steps": 16, "loop": "c"}),
    (3, (300, 700), "float64", "uniform", True, {"max_pass_steps": 16}),
])
def test_gpu_slabs_match_single_engine_and_oracle(tmp_path, world, shape, dtype, materials, overlap,
                                                  options):
    import fdtd2d_amd as fd
    from oracle import fdtd_numpy as onp
    r, c = shape
    options = dict(options or {})
    loop = options.pop("loop", "python")
    expect_overlap = options.pop("expect_overlap", None)
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
               extent=(options or {}).get("extent"), expect_overlap=expect_overlap)
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


# ---- 8 ranks: BASELINE configs[3] / configs[4]'s decomposition, one thread per rank (tests/local_slabs.py) -------

def _slab_state(r, c, n, seed, eps_hi=10.0, zero_fields=False):
    from oracle import fdtd_numpy as onp
    rng = np.random.default_rng(seed)
    st = dict(Ez=rng.standard_normal((r, c)), Hx=rng.standard_normal((r, c - 1)) * 1e-3,
              Hy=rng.standard_normal((r - 1, c)) * 1e-3,
              eps=onp.EPS0 * rng.uniform(1, eps_hi, (r, c)), mu=onp.MU0 * np.ones((r, c)),
              amps=rng.standard_normal(n), dt=DT, dx=DX)
    if zero_fields:
        for k in ("Ez", "Hx", "Hy"):
            st[k][:] = 0
    st["eps"][0, 0] = onp.EPS0
    return st


@pytest.mark.parametrize("world,shape,cycle_opt,materials,n", [
    (8, (8 * 48, 520), 16, "uniform", 45),     # overlapped 16-step cycles, 16 + 16 + 13
    (8, (8 * 40, 300), None, "array", 29),     # small slabs: 8-step cycles
    (5, (5 * 37, 700), 16, "array", 35),       # the shortest slabs that still overlap with 16-step cycles
])
def test_eight_mur_slabs_in_one_process_match_single_engine_and_oracle(world, shape, cycle_opt, materials, n):
    import fdtd2d_amd as fd
    from oracle import fdtd_numpy as onp
    from local_slabs import run_local_slabs
    r, c = shape
    st = _slab_state(r, c, n, r + c)
    if materials == "uniform":
        st["eps"][:] = 2.5 * onp.EPS0
    src = (r // world, c // 2)
    got, cycle, overlapped, launches = run_local_slabs(fd, world, shape, np.float32, "mur", st, n, src,
                                                       cycle_opt=cycle_opt, materials=materials)
    assert cycle == (16 if cycle_opt else 8) and overlapped and min(launches) > 0
    ref = [st[k].astype(np.float32) for k in ("Ez", "Hx", "Hy")]
    onp.leapfrog(*ref, st["eps"].astype(np.float32), st["mu"].astype(np.float32), DT, DX, n, src[0], src[1],
                 amps=st["amps"])
    for a, b, k in zip(got, ref, ("Ez", "Hx", "Hy")):
        assert np.array_equal(a, b), f"{k}: {world} slabs differ from the oracle at {np.argwhere(a != b)[:3]}"


@pytest.mark.parametrize("cycle_opt,n,overlap", [(16, 45, True), (16, 37, False), (None, 21, True)])
def test_eight_pml_slabs_in_one_process_match_single_engine_and_oracle(cycle_opt, n, overlap):
    """BASELINE configs[4]'s decomposition -- 8 row slabs, boundary="pml", float32 -- on a small grid: ranks 0 and 7
    hold the top / bottom layer, every rank the two column layers; 4-field halo messages, overlapped 16-step cycles
    on the level-split pair (k_bulk_split + k_bulk_split_pml), the run loop in C.  Equal to the single engine and
    to the build's PML oracle (parity unpinned: no time-domain PML in the reference) bit for bit."""
    import fdtd2d_amd as fd
    from oracle import pml_numpy as pm
    from local_slabs import run_local_slabs
    world, r, c = 8, 8 * 64, 600
    st = _slab_state(r, c, n, 77 + n, eps_hi=3.0, zero_fields=True)
    src = (2 * 64, 301)                                  # on a cut
    got, cycle, overlapped, launches = run_local_slabs(fd, world, (r, c), np.float32, "pml", st, n, src,
                                                       cycle_opt=cycle_opt, overlap=overlap)
    assert cycle == (16 if cycle_opt else 8) and overlapped == overlap and min(launches) > 0
    eps, mu = st["eps"].astype(np.float32), st["mu"].astype(np.float32)
    with fd.Engine(r, c, DT, DX, dtype=np.float32, boundary="pml") as eng:
        eng.set_materials(eps, mu).set_pml()
        if cycle_opt:
            eng.set_option(max_pass_steps=cycle_opt)
        eng.run(n, src[0], src[1], st["amps"])
        one = eng.download()
    S = (1 / np.sqrt(fd.EPS0 * fd.MU0) * DT) / DX
    ref = [np.zeros((r, c), np.float32), np.zeros((r, c), np.float32), np.zeros((r, c - 1), np.float32),
           np.zeros((r - 1, c), np.float32)]
    pm.leapfrog(*ref, eps, mu, DT, DX, n, src[0], src[1], st["amps"], pm.profiles(r, c, S, dtype=np.float32))
    assert np.abs(one[0]).max() > 0
    for a, b, c_, k in zip(got, one, (ref[0], ref[2], ref[3]), ("Ez", "Hx", "Hy")):
        assert np.array_equal(a, b), f"{k}: 8 slabs differ from the single engine at {np.argwhere(a != b)[:3]}"
        assert np.array_equal(a, c_), f"{k}: 8 slabs differ from the PML oracle at {np.argwhere(a != c_)[:3]}"


def test_pml_factor_arrays_without_unit_structure_keep_the_8_step_cycle():
    """Round-2 advisor finding: cycle_steps() said 16 for every float32 PML handle with uniform mu, also when the
    factor arrays are not 1 outside the layer -- then only the 8-step kernel can run them, and a slab loop that
    took 16 as its exchange cycle failed mid-run.  Now the handle reports 8, the slabs cycle by 8 and the result
    equals the single engine and the oracle run with the same arrays."""
    import fdtd2d_amd as fd
    from oracle import pml_numpy as pm
    from local_slabs import run_local_slabs
    world, r, c, n = 3, 3 * 70, 300, 27
    st = _slab_state(r, c, n, 5, eps_hi=3.0, zero_fields=True)
    S = (1 / np.sqrt(fd.EPS0 * fd.MU0) * DT) / DX
    P = pm.profiles(r, c, S, dtype=np.float32)
    P["ahr"] = P["ahr"].copy()
    P["ahr"][r // 2] = np.float32(0.97)                  # an H factor != 1 outside the layer
    src = (70, 150)
    got, cycle, overlapped, _ = run_local_slabs(fd, world, (r, c), np.float32, "pml", st, n, src, cycle_opt=16,
                                                pml=dict(profiles=P))
    assert cycle == 8 and overlapped
    eps, mu = st["eps"].astype(np.float32), st["mu"].astype(np.float32)
    with fd.Engine(r, c, DT, DX, dtype=np.float32, boundary="pml") as eng:
        eng.set_materials(eps, mu).set_pml(profiles=P).set_option(max_pass_steps=16)
        assert eng.cycle_steps == 8
        eng.run(n, src[0], src[1], st["amps"])
        one = eng.download()
        assert eng.info(16) > 0
    ref = [np.zeros((r, c), np.float32), np.zeros((r, c), np.float32), np.zeros((r, c - 1), np.float32),
           np.zeros((r - 1, c), np.float32)]
    pm.leapfrog(*ref, eps, mu, DT, DX, n, src[0], src[1], st["amps"], P)
    for a, b, c_, k in zip(got, one, (ref[0], ref[2], ref[3]), ("Ez", "Hx", "Hy")):
        assert np.array_equal(a, b), f"{k}: slabs differ from the single engine"
        assert np.array_equal(a, c_), f"{k}: slabs differ from the PML oracle"


def test_c_loop_refuses_overlap_on_slabs_that_cut_the_zone():
    """fdtd2d_run_slab(overlap=1) on 35-row slabs with 16-step cycles: the first / last rank returns FDTD2D_E_ARG
    before it posts anything (its edge piece would cut through the 21-row zone)."""
    import fdtd2d_amd as fd
    from fdtd2d_amd import _abi
    r, c = 105, 300
    with fd.Engine(r, c, DT, DX, dtype=np.float32, slab=(0, 35, 16)) as eng:
        eng.set_materials().set_option(max_pass_steps=16)
        import hipmem
        send, recv = hipmem.DevBuf(eng.halo_bytes), hipmem.DevBuf(eng.halo_bytes)
        called = []
        eng.slab_attach({1: (send.ptr, recv.ptr)}, lambda *a: called.append(a) or 0)
        with pytest.raises(_abi.Fdtd2dError) as e:
            eng.run_slab(16, 16, True)
        assert e.value.code == _abi.E_ARG and not called


@pytest.mark.parametrize("side,xcd,materials", [(2, 0, "uniform"), (2, 1, "array"), (4, 1, "uniform")])
def test_slabs_with_strips_of_several_waves(side, xcd, materials):
    """The launch shapes round 3 added -- 2 / 4 waves side by side per level group, tasks dealt out XCD by XCD -- inside a
    row-slab run (pieces of a pass next to the cuts, interior piece, overlapped 16-step cycles): 3 slabs of a 240-row
    grid wide enough for them, equal to the oracle bit for bit."""
    import fdtd2d_amd as fd
    from oracle import fdtd_numpy as onp
    from local_slabs import run_local_slabs
    world, r, c, n = 3, 240, 4200 * (2 if side == 4 else 1), 35
    st = _slab_state(r, c, n, side + xcd)
    if materials == "uniform":
        st["eps"][:] = 2.5 * onp.EPS0
    src = (80, 251)
    got, cycle, overlapped, launches = run_local_slabs(fd, world, (r, c), np.float32, "mur", st, n, src, cycle_opt=16,
                                                       materials=materials, options=dict(side_waves=side, xcd_map=xcd))
    assert cycle == 16 and overlapped and min(launches) > 0
    ref = [st[k].astype(np.float32) for k in ("Ez", "Hx", "Hy")]
    onp.leapfrog(*ref, st["eps"].astype(np.float32), st["mu"].astype(np.float32), DT, DX, n, src[0], src[1], amps=st["amps"])
    for a, b, k in zip(got, ref, ("Ez", "Hx", "Hy")):
        assert np.array_equal(a, b), f"{k}: slabs (side {side}, xcd {xcd}) differ from the oracle at {np.argwhere(a != b)[:3]}"
