"""Pins the CPU oracle (NumPy and C restatements) to vectors produced by RUNNING
the reference (tests/golden/make_golden.py).  Everything here is bit-for-bit:
`np.array_equal` on every field, for the reference's float64 default and for the
reference handed float32 arrays.  CPU only."""
import os

import numpy as np
import pytest

from oracle import c_oracle
from oracle import fdtd_numpy as onp

DTYPES = [("f64", np.float64), ("f32", np.float32)]
IMPLS = [("numpy", onp), ("c", c_oracle)]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("impl_name,impl", IMPLS)
@pytest.mark.parametrize("tag,dtype", DTYPES)
@pytest.mark.parametrize("name", ["g1_single_48x40", "g6_single_11x11", "g6_single_12x13"])
def test_half_steps_in_isolation(golden_dir, name, tag, dtype, impl_name, impl):
    g = _load(golden_dir, name)
    dt, dx = float(g["dt"]), float(g["dx"])
    f = lambda k: np.ascontiguousarray(g[k].astype(dtype))
    Ez, Hx, Hy, eps, mu = f("Ez"), f("Hx"), f("Hy"), f("eps"), f("mu")
    # H half-step alone
    hx, hy = impl.update_h(Ez, Hx, Hy, mu, eps, dt, dx)
    assert hx is Hx and hy is Hy, "H update must work in place and return the same objects"
    assert np.array_equal(Hx, g[f"h_Hx_{tag}"]) and np.array_equal(Hy, g[f"h_Hy_{tag}"])
    # E half-step alone, from the original H
    Ez2 = f("Ez")
    out = impl.update_e(Ez2, f("Hx"), f("Hy"), mu, eps, dt, dx)
    assert out is Ez2
    assert np.array_equal(Ez2, g[f"e_Ez_{tag}"])
    # full H -> E step
    impl.update_e(Ez, Hx, Hy, mu, eps, dt, dx)
    assert np.array_equal(Ez, g[f"step_Ez_{tag}"])
    assert Ez.dtype == dtype


def _materials(g, dtype):
    r, c = int(g["rows"]), int(g["cols"])
    if "eps" in g.files:
        eps = g["eps"].astype(dtype)
    else:
        eps = np.full((r, c), float(g["eps_uniform"])).astype(dtype)
    mu = np.full((r, c), onp.MU0).astype(dtype)
    return np.ascontiguousarray(eps), np.ascontiguousarray(mu)


@pytest.mark.parametrize("impl_name", ["numpy", "c"])
@pytest.mark.parametrize("tag,dtype", DTYPES)
@pytest.mark.parametrize("name", ["g2_vacuum_64x64", "g3_disk_64x80", "g4_config1_256x256",
                                  "g7_vacuum_96x96_2000"])
def test_time_loop(golden_dir, name, tag, dtype, impl_name):
    g = _load(golden_dir, name)
    r, c, n = int(g["rows"]), int(g["cols"]), int(g["nsteps"])
    dt, dx = float(g["dt"]), float(g["dx"])
    sr, sc = (int(v) for v in g["src"])
    eps, mu = _materials(g, dtype)
    Ez, Hx, Hy = onp.grid_zeros(r, c, dtype)
    amps = g["amps"]
    snaps = [int(s) for s in g["snaps"] if f"Ez_{tag}_{int(s)}" in g.files]
    done = 0
    for s in snaps:
        if impl_name == "numpy":
            onp.leapfrog(Ez, Hx, Hy, eps, mu, dt, dx, s - done, sr, sc, amps=amps[done:s])
        else:
            c_oracle.run(Ez, Hx, Hy, eps, mu, dt, dx, s - done, sr, sc, amps=amps[done:s])
        done = s
        for a, k in ((Ez, "Ez"), (Hx, "Hx"), (Hy, "Hy")):
            assert np.array_equal(a, g[f"{k}_{tag}_{s}"]), f"{k} differs at step {s}"
    assert done == n


def test_numpy_oracle_own_ricker_matches_reference_amplitudes(golden_dir):
    """The oracle's waveform (not the stored amps) reproduces the reference's values."""
    g = _load(golden_dir, "g2_vacuum_64x64")
    dt, fc = float(g["dt"]), float(g["fc"])
    mine = np.array([onp.ricker_amplitude(i * dt, fc) for i in range(int(g["nsteps"]))])
    # same NumPy on the same machine is bit-identical; another libm may differ in the last ulp
    np.testing.assert_allclose(mine, g["amps"], rtol=4e-16, atol=1e-300)


def test_scalars(golden_dir):
    g = _load(golden_dir, "g5_scalars")
    dt, fc = float(g["dt"]), float(g["fc"])
    for i, r, s in zip(g["steps"], g["ricker"], g["sinusoidal"]):
        np.testing.assert_allclose(onp.ricker_amplitude(int(i) * dt, fc), r, rtol=4e-16)
        np.testing.assert_allclose(c_oracle.ricker(int(i) * dt, fc), r, rtol=1e-14)
        np.testing.assert_allclose(onp.sinusoidal_amplitude(int(i) * dt, fc), s, rtol=4e-16,
                                   atol=1e-300)
    assert onp.EPS0 == float(g["eps_vac"]) and onp.MU0 == float(g["mu_vac"])
    shapes = [a.shape for a in onp.grid_zeros(7, 9)]
    assert [tuple(s) for s in g["grid_shapes"]] == shapes
    assert str(g["grid_dtype"]) == "float64"


def test_known_answers_from_survey():
    """Constants recorded by probing the reference."""
    dt, dx, fc = 5e-14, 1e-4, 30e9
    assert onp.ricker_amplitude(0 * dt, fc) == pytest.approx(-0.0009692515861872089, rel=1e-14)
    assert onp.ricker_amplitude(666 * dt, fc) == pytest.approx(0.9999703914303184, rel=1e-14)
    assert dt / (onp.EPS0 * dx) == pytest.approx(56.47050319735989, rel=1e-15)
    assert dt / (onp.MU0 * dx) == pytest.approx(0.0003978873577297383, rel=1e-15)
    assert onp.mur_coefficient(onp.MU0, onp.EPS0, dt, dx) == pytest.approx(-0.7392872804216724, rel=1e-14)
    eps, mu = onp.vacuum_materials(4, 4)
    assert onp.courant_number(eps, mu, dt, dx) == pytest.approx(0.14989629517391773, rel=1e-14)


def test_config1_end_state(golden_dir):
    g = _load(golden_dir, "g4_config1_256x256")
    Ez, Hx, Hy = g["Ez_f64_500"], g["Hx_f64_500"], g["Hy_f64_500"]
    assert np.abs(Ez).max() == 0.17954891765410197
    assert Ez.sum() == pytest.approx(-89.84260278212417, rel=1e-12)
    assert np.abs(Hx).max() == 0.0006013684170245006
    assert np.abs(Hy).max() == 0.0006013684170245006


def test_staged_numpy_oracle_rejects_tiny_grids():
    Ez, Hx, Hy = onp.grid_zeros(10, 12)
    eps, mu = onp.vacuum_materials(10, 12)
    with pytest.raises(ValueError):
        onp.update_e(Ez, Hx, Hy, mu, eps, 5e-14, 1e-4)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_c_and_numpy_oracles_agree_on_random_state(dtype):
    rng = np.random.default_rng(7)
    r, c = 37, 53
    mk = lambda *s: np.ascontiguousarray(rng.standard_normal(s).astype(dtype))
    Ez, Hx, Hy = mk(r, c), mk(r, c - 1) * dtype(1e-3), mk(r - 1, c) * dtype(1e-3)
    eps = (onp.EPS0 * rng.uniform(1, 12, (r, c))).astype(dtype)
    mu = (onp.MU0 * rng.uniform(1, 3, (r, c))).astype(dtype)
    a = [x.copy() for x in (Ez, Hx, Hy)]
    b = [x.copy() for x in (Ez, Hx, Hy)]
    for n in range(20):
        onp.update_h(a[0], a[1], a[2], mu, eps, 5e-14, 1e-4)
        onp.update_e(a[0], a[1], a[2], mu, eps, 5e-14, 1e-4)
        onp.add_point(a[0], 9, 11, 0.25 * n)
    c_oracle.run(b[0], b[1], b[2], eps, mu, 5e-14, 1e-4, 20, 9, 11,
                 amps=np.array([0.25 * n for n in range(20)]))
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
