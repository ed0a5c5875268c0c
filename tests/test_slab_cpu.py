"""CPU (gloo, world_size 2 and 3): the row-slab runner -- plan, halo exchange order, source
handling, chunked runs -- reproduces the single-domain oracle bit for bit.  The per-rank
engine is the oracle-backed stand-in of tests/fake_engine.py; on the GPU box the same
harness runs with the real HIP engine (tests/test_gpu_slab.py)."""
import os

import numpy as np
import pytest

from oracle import fdtd_numpy as onp
from fdtd2d_amd.slab import plan_slabs

from dist_harness import run_job

DT, DX = 5e-14, 1e-4


def test_plan_slabs():
    assert plan_slabs(100, 1) == [(0, 100)]
    assert plan_slabs(64, 2) == [(0, 32), (32, 64)]
    p = plan_slabs(103, 4)
    assert p[0][0] == 0 and p[-1][1] == 103 and all(a[1] == b[0] for a, b in zip(p, p[1:]))
    assert max(b - a for a, b in p) - min(b - a for a, b in p) <= 1
    assert plan_slabs(16384, 4) == [(0, 4096), (4096, 8192), (8192, 12288), (12288, 16384)]
    with pytest.raises(ValueError):
        plan_slabs(40, 4, halo=8)  # 10-row slabs cannot hold the 6+8 row clearance


def _state(tmp_path, r, c, seed, nsteps, vary_mu=False):
    rng = np.random.default_rng(seed)
    st = dict(Ez=rng.standard_normal((r, c)), Hx=rng.standard_normal((r, c - 1)) * 1e-3,
              Hy=rng.standard_normal((r - 1, c)) * 1e-3,
              eps=onp.EPS0 * rng.uniform(1, 10, (r, c)),
              mu=onp.MU0 * (rng.uniform(1, 3, (r, c)) if vary_mu else np.ones((r, c))),
              amps=rng.standard_normal(nsteps))
    path = os.path.join(tmp_path, "state.npz")
    np.savez(path, **st)
    return st, path


@pytest.mark.parametrize("world,shape,src", [(2, (40, 24), (19, 5)), (2, (41, 30), (20, 29)),
                                             (3, (66, 20), (22, 3)), (3, (70, 25), (1, 1))])
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("overlap", [True, False])
def test_slab_runner_matches_single_domain_oracle(tmp_path, world, shape, src, dtype, overlap):
    """27 steps in chunks (8+8+3, then 8): source on a cut / in a halo / on the frame."""
    r, c = shape
    st, path = _state(str(tmp_path), r, c, 11 * r + c, 27, vary_mu=True)
    job = dict(engine="fake", shape=shape, dtype=dtype, dt=DT, dx=DX, state=path, src=src,
               chunks=[19, 8], materials="array", overlap=overlap)
    got = run_job(world, job, str(tmp_path))
    dt_ = np.dtype(dtype)
    ref = [st[k].astype(dt_) for k in ("Ez", "Hx", "Hy")]
    onp.leapfrog(*ref, st["eps"].astype(dt_), st["mu"].astype(dt_), DT, DX, 27, src[0], src[1],
                 amps=st["amps"])
    for a, b, k in zip(got, ref, ("Ez", "Hx", "Hy")):
        assert a.shape == b.shape and np.array_equal(a, b), k


@pytest.mark.parametrize("cycle", [8, 16])
@pytest.mark.parametrize("overlap", [True, False])
def test_slab_runner_16_row_halo(tmp_path, cycle, overlap):
    """Slabs tall enough for the 16-row halo: 16 steps per exchange where the engine runs
    16-step passes (cycle 16), two 8-step cycles' worth of rows per message otherwise."""
    r, c = 50, 24
    st, path = _state(str(tmp_path), r, c, 77, 51, vary_mu=True)
    job = dict(engine="fake", shape=(r, c), dtype="float32", dt=DT, dx=DX, state=path,
               src=(25, 3), chunks=[35, 16], materials="array", overlap=overlap, cycle=cycle)
    got = run_job(2, job, str(tmp_path))
    ref = [st[k].astype(np.float32) for k in ("Ez", "Hx", "Hy")]
    onp.leapfrog(*ref, st["eps"].astype(np.float32), st["mu"].astype(np.float32), DT, DX, 51,
                 25, 3, amps=st["amps"])
    for a, b, k in zip(got, ref, ("Ez", "Hx", "Hy")):
        assert np.array_equal(a, b), k


def test_slab_runner_line_source_across_cut(tmp_path):
    """A column line source that crosses the cut between two slabs (and their halos)."""
    r, c = 48, 22
    st, path = _state(str(tmp_path), r, c, 9, 19, vary_mu=True)
    job = dict(engine="fake", shape=(r, c), dtype="float32", dt=DT, dx=DX, state=path,
               src=(10, 7), chunks=[19], materials="array", extent=(30, 2))
    got = run_job(2, job, str(tmp_path))
    ref = [st[k].astype(np.float32) for k in ("Ez", "Hx", "Hy")]
    onp.leapfrog(*ref, st["eps"].astype(np.float32), st["mu"].astype(np.float32), DT, DX, 19,
                 10, 7, amps=st["amps"], extent=(30, 2))
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)


def test_slab_runner_uniform_materials(tmp_path):
    r, c = 48, 22
    st, path = _state(str(tmp_path), r, c, 5, 16)
    st["eps"][:] = 2 * onp.EPS0
    st["mu"][:] = onp.MU0
    np.savez(path, **st)
    job = dict(engine="fake", shape=(r, c), dtype="float32", dt=DT, dx=DX, state=path,
               src=(24, 11), chunks=[16], materials="uniform")
    got = run_job(2, job, str(tmp_path))
    ref = [st[k].astype(np.float32) for k in ("Ez", "Hx", "Hy")]
    onp.leapfrog(*ref, st["eps"].astype(np.float32), st["mu"].astype(np.float32), DT, DX, 16,
                 24, 11, amps=st["amps"])
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)


def _check(got, st, dtype, nsteps, src):
    dt_ = np.dtype(dtype)
    ref = [st[k].astype(dt_) for k in ("Ez", "Hx", "Hy")]
    onp.leapfrog(*ref, st["eps"].astype(dt_), st["mu"].astype(dt_), DT, DX, nsteps, src[0], src[1],
                 amps=st["amps"])
    for a, b, k in zip(got, ref, ("Ez", "Hx", "Hy")):
        assert a.shape == b.shape and np.array_equal(a, b), k


@pytest.mark.parametrize("rows,overlapped", [(111, True), (110, False)])
def test_three_ranks_agree_on_the_exchange_mode(tmp_path, rows, overlapped):
    """111 rows = three 37-row slabs: every rank may overlap (2*16+5 rows: two 16-row edge pieces and, on the first /
    last rank, an interior piece that holds the whole 21-row zone).  110 rows = 37/37/36:
    the short slab cannot, so NO rank does -- a per-rank decision would leave ranks 0 and 1 in the
    overlapped cycle (send after the pass) and rank 2 in the plain one (send before it): deadlock."""
    c, n = 18, 40
    st, path = _state(str(tmp_path), rows, c, rows, n, vary_mu=True)
    job = dict(engine="fake", shape=(rows, c), dtype="float32", dt=DT, dx=DX, state=path, src=(40, 5),
               chunks=[32, 8], materials="array", overlap=True, cycle=16, expect_overlap=overlapped,
               expect_cycle=16)
    _check(run_job(3, job, str(tmp_path)), st, "float32", n, (40, 5))


def test_ranks_whose_engines_disagree_use_the_shortest_cycle(tmp_path):
    """An engine chooses 16- or 8-step passes from its own slab size; slabs differ by a row, so the
    answers can differ.  The runner takes the minimum and hands it back to every engine."""
    rows, c, n = 99, 16, 24
    st, path = _state(str(tmp_path), rows, c, 3, n, vary_mu=True)
    job = dict(engine="fake", shape=(rows, c), dtype="float32", dt=DT, dx=DX, state=path, src=(50, 4),
               chunks=[24], materials="array", overlap=True, cycle=[16, 16, 8], expect_cycle=8)
    _check(run_job(3, job, str(tmp_path)), st, "float32", n, (50, 4))


@pytest.mark.parametrize("world,rows", [(4, 152), (8, 304), (8, 180)])
def test_four_and_eight_slabs(tmp_path, world, rows):
    """BASELINE configs 4 and 5 decompose into 4 and 8 slabs: interior ranks have two neighbours,
    the source sits on a cut, 38-row slabs overlap (16-step cycles), 22-row slabs do not."""
    c, n = 14, 35
    st, path = _state(str(tmp_path), rows, c, world * rows, n, vary_mu=True)
    src = (rows // 2, 6)
    tall = rows // world >= 37
    job = dict(engine="fake", shape=(rows, c), dtype="float32", dt=DT, dx=DX, state=path, src=src,
               chunks=[16, 19], materials="array", overlap=True, cycle=16, expect_overlap=tall)
    _check(run_job(world, job, str(tmp_path)), st, "float32", n, src)
