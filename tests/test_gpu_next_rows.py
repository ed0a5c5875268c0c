"""GPU: device-side consumers of Ez next to the loop."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DT, DX, FC = 5e-14, 1e-4, 30e9


@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
def test_device_snapshot_equals_reference_png(golden_dir, tag, dtype):
    """Engine state = the golden mid-run field; the device-computed colour indices rendered
    through the host table give the reference's PNG pixels exactly."""
    import fdtd2d_amd as fd
    g3 = np.load(os.path.join(golden_dir, "g3_disk_64x80.npz"))
    g = np.load(os.path.join(golden_dir, f"n1_snapshot_{tag}.npz"))
    Ez, Hx, Hy = (g3[f"{k}_{tag}_400"] for k in ("Ez", "Hx", "Hy"))
    with fd.Engine(64, 80, DT, DX, dtype=dtype) as eng:
        eng.set_materials(g3["eps"].astype(dtype), np.full((64, 80), fd.MU0, dtype))
        eng.upload(Ez, Hx, Hy)
        img = fd.capture_snapshot(eng, g3["eps"], None, float(g["vmax"]), float(g["vmin"]))
        assert np.array_equal(img, g["rgb"])
        idx = eng.snapshot_index(-0.2, 0.2, 1)
        assert np.array_equal(idx, fd.snapshot_indices(Ez, 0.2, -0.2))
        for stride in (2, 3, 7):
            dec = eng.snapshot_index(-0.2, 0.2, stride)
            assert np.array_equal(dec, idx[::stride, ::stride])


def test_snapshot_cadence_in_run_fdtd_matches_reference_loop(golden_dir):
    """on_frame fires after step i when i % (nsteps // nframes) == 0, as fdtd.py:36-38."""
    import fdtd2d_amd as fd
    from oracle import fdtd_numpy as onp
    seen = []
    Ez, Hx, Hy = fd.run_fdtd(48, 56, DT, DX, 50, dtype=np.float64, nframes=10,
                             on_frame=lambda i, E: seen.append((i, E.copy())))
    assert [i for i, _ in seen] == list(range(0, 50, 5))
    ref = onp.grid_zeros(48, 56)
    eps, mu = onp.vacuum_materials(48, 56)
    for i in range(50):
        onp.leapfrog(*ref, eps, mu, DT, DX, 1, 24, 28, step0=i)
        if i % 5 == 0:
            assert np.array_equal(seen[i // 5][1], ref[0]), i
    assert np.array_equal(Ez, ref[0])


def test_sinusoidal_source_run(golden_dir):
    """N3: the CW source of main.py:190-195 through run_fdtd, against the oracle."""
    import fdtd2d_amd as fd
    from oracle import fdtd_numpy as onp
    n = 64
    Ez, Hx, Hy = fd.run_fdtd(64, 64, DT, DX, n, source=("sinusoidal", 20, 30, FC), dtype=np.float32)
    ref = onp.grid_zeros(64, 64, np.float32)
    eps, mu = onp.vacuum_materials(64, 64, np.float32)
    amps = [onp.sinusoidal_amplitude(i * DT, FC) for i in range(n)]
    onp.leapfrog(*ref, eps, mu, DT, DX, n, 20, 30, amps=amps)
    assert np.abs(ref[0]).max() > 0.1
    for a, b in zip((Ez, Hx, Hy), ref):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_device_reductions(dtype):
    """N4: sum of squares / max |.| reduced on the device (float64 accumulation)."""
    import fdtd2d_amd as fd
    rng = np.random.default_rng(1)
    r, c = 300, 517
    Ez = rng.standard_normal((r, c)).astype(dtype)
    Hx = rng.standard_normal((r, c - 1)).astype(dtype)
    Hy = rng.standard_normal((r - 1, c)).astype(dtype)
    with fd.Engine(r, c, DT, DX, dtype=dtype) as eng:
        eng.set_materials()
        eng.upload(Ez, Hx, Hy)
        for name, a in (("Ez", Ez), ("Hx", Hx), ("Hy", Hy)):
            s, m = eng.reduce(name)
            assert m == np.abs(a).max()
            assert s == pytest.approx(np.sum(a.astype(np.float64) ** 2), rel=1e-12)


def test_time_launches_and_counters():
    """Measurement helpers of the C ABI: per-launch HIP-event timing and launch counters."""
    import fdtd2d_amd as fd
    with fd.Engine(512, 512, DT, DX, dtype=np.float32) as eng:
        eng.set_materials()
        p0 = eng.info(16)
        ms = eng.time_launches(6, 8)
        assert ms.shape == (6,) and np.all(ms > 0) and np.all(ms < 50)
        assert eng.info(16) - p0 == 6 and eng.step_count == 48
        assert eng.bytes_per_cell_step == 24
        eng.timer_start()
        eng.run(16)
        assert 0 < eng.timer_stop() < 100


def test_reference_experiment_script(tmp_path, golden_dir):
    """`python -m fdtd2d_amd.fdtd` with the reference's fdtd.py defaults scaled down: frames
    at the reference's cadence, final field equal to the oracle's loop."""
    from fdtd2d_amd import fdtd as script
    from oracle import fdtd_numpy as onp
    frames = os.path.join(str(tmp_path), "frames")
    Ez, Hx, Hy = script.main(["--rows", "64", "--cols", "72", "--nsteps", "40", "--nframes", "8",
                              "--frames", frames])
    assert sorted(os.listdir(frames)) == [f"frame_{k:04d}.png" for k in range(8)]
    ref = onp.grid_zeros(64, 72)
    eps, mu = onp.vacuum_materials(64, 72)
    onp.leapfrog(*ref, eps, mu, DT, DX, 40, 32, 36)
    assert np.array_equal(Ez, ref[0]) and np.array_equal(Hx, ref[1])


@pytest.mark.parametrize("tag,dtype", [("f32", np.float32), ("f64", np.float64)])
@pytest.mark.parametrize("probe", [(70, 250), (0, 0), (3, 466), (129, 235), (64, 2), (20, 236), (126, 4)])
@pytest.mark.parametrize("max_steps", [0, 8, 16])
def test_probe_time_series_matches_oracle(tag, dtype, probe, max_steps):
    """N4: Ez at one cell after every step -- interior, corners, side bands, zones, a strip seam --
    including the steps inside 8- and 16-step passes (recomputed by the probe tile), against
    the oracle's per-step values; 45 steps with a patch source, array eps."""
    import fdtd2d_amd as fd
    from oracle import fdtd_numpy as onp
    # (float64 with a probe set: the 16-step kernel has no probe tile in float64 -- the engine runs 8-step passes)
    r, c, n = 130, 470, 45
    rng = np.random.default_rng(probe[0] * 7 + probe[1])
    Ez = rng.standard_normal((r, c)).astype(dtype)
    Hx = (rng.standard_normal((r, c - 1)) * 1e-3).astype(dtype)
    Hy = (rng.standard_normal((r - 1, c)) * 1e-3).astype(dtype)
    eps = (onp.EPS0 * rng.uniform(1, 10, (r, c))).astype(dtype)
    mu = np.full((r, c), onp.MU0, dtype)
    amps = rng.standard_normal(n)
    want = []
    ref = [a.copy() for a in (Ez, Hx, Hy)]
    onp.leapfrog(*ref, eps, mu, 5e-14, 1e-4, n, 60, 230, amps=amps, extent=(4, 12),
                 on_step=lambda i, E, *_: want.append(float(E[probe])))
    with fd.Engine(r, c, 5e-14, 1e-4, dtype=dtype) as eng:
        eng.set_materials(eps, mu).set_option(max_pass_steps=max_steps).set_source_extent(4, 12)
        eng.upload(Ez, Hx, Hy)
        eng.run(3, 60, 230, amps[:3])                     # the probe starts mid-run
        eng.set_probe(probe[0], probe[1], n)
        eng.run(n - 3, 60, 230, amps[3:])
        got = eng.read_probe()
        fields = eng.download()
        assert eng.read_probe(5, 4).tolist() == got[5:9].tolist()
        with pytest.raises(fd.Fdtd2dError):
            eng.read_probe(0, n + 1)
    assert got.shape == (n - 3,) and np.array_equal(got, np.array(want[3:]))
    for a, b in zip(fields, ref):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("tag,dtype", [("f32", np.float32), ("f64", np.float64)])
@pytest.mark.parametrize("every,max_steps", [(16, 16), (4, 16), (7, 8), (1, 16), (16, 0)])
def test_running_fourier_transform_matches_oracle(tag, dtype, every, max_steps):
    """N4: the running transform of Ez at two frequencies over a window, sampled every `every` steps (the loop cuts its
    passes at the sampled steps: 16-step passes for every = 16, short passes of 4 and 7 levels, single steps), against the
    same sum over the oracle's Ez sequence in float64: relative 1e-12 (the only difference is libm's cos / sin vs NumPy's).
    The fields themselves stay value-identical."""
    import fdtd2d_amd as fd
    from oracle import fdtd_numpy as onp
    r, c, n = 150, 300, 80
    rng = np.random.default_rng(every * 10 + max_steps)
    Ez = rng.standard_normal((r, c)).astype(dtype)
    Hx = (rng.standard_normal((r, c - 1)) * 1e-3).astype(dtype)
    Hy = (rng.standard_normal((r - 1, c)) * 1e-3).astype(dtype)
    eps = (onp.EPS0 * rng.uniform(1, 10, (r, c))).astype(dtype)
    mu = np.full((r, c), onp.MU0, dtype)
    amps = rng.standard_normal(n)
    win = (3, 250, 60, 45)                                # reaches into the top zone and across a strip seam
    om = 2 * np.pi * np.array([30e9, 47e9])
    want = np.zeros((2, win[2], win[3]), np.complex128)

    def on_step(i, E, *_):
        k = i + 1
        if k % every == 0:
            w = E[win[0]:win[0] + win[2], win[1]:win[1] + win[3]].astype(np.float64)
            for f in range(2):
                want[f] += w * np.cos(om[f] * (k * 5e-14)) + 1j * (w * -np.sin(om[f] * (k * 5e-14)))
    ref = [a.copy() for a in (Ez, Hx, Hy)]
    onp.leapfrog(*ref, eps, mu, 5e-14, 1e-4, n, 70, 260, amps=amps, on_step=on_step)
    with fd.Engine(r, c, 5e-14, 1e-4, dtype=dtype) as eng:
        eng.set_materials(eps, mu).set_option(max_pass_steps=max_steps)
        eng.upload(Ez, Hx, Hy)
        eng.set_dft(win, om, every)
        eng.run(50, 70, 260, amps[:50])
        eng.run(30, 70, 260, amps[50:])
        got = eng.read_dft()
        fields = eng.download()
        with pytest.raises(fd.Fdtd2dError):
            eng.set_dft((140, 0, 20, 10), om, 4)            # window leaves the grid
        eng.set_dft(win, (), 1)                             # removes it
    for a, b in zip(fields, ref):
        assert np.array_equal(a, b)
    assert got.shape == want.shape and np.abs(want).max() > 0
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max(), np.abs(got - want).max() / np.abs(want).max()


def test_running_fourier_transform_on_slabs_needs_an_aligned_cycle():
    """A slab handle whose passes are issued in pieces: a 16-step pass may end on a sampled step (every = 16) but not run
    over one (every = 10) -- the commit says so instead of silently losing samples; fdtd2d_run_slab refuses up front."""
    import fdtd2d_amd as fd
    r, c = 200, 300
    om = 2 * np.pi * np.array([30e9])
    with fd.Engine(r, c, 5e-14, 1e-4, dtype=np.float32, slab=(0, 100, 16)) as eng:
        eng.set_materials().set_option(max_pass_steps=16)
        eng.set_dft((10, 10, 50, 50), om, 16)
        eng.pass_rows(16, 0, 100)              # rows [0, 100): the owned rows of this top slab (halo rows stay stale)
        eng.pass_commit()
        assert eng.step_count == 16 and np.abs(eng.read_dft()).max() == 0.0      # zero fields: a sample of zeros
    with fd.Engine(r, c, 5e-14, 1e-4, dtype=np.float32, slab=(0, 100, 16)) as eng:
        eng.set_materials().set_option(max_pass_steps=16)
        eng.set_dft((10, 10, 50, 50), om, 10)
        eng.pass_rows(16, 0, 100)
        with pytest.raises(fd.Fdtd2dError):
            eng.pass_commit()
