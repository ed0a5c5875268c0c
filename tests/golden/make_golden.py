#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Build-container only: needs /root/reference (read-only checkout of
skunnavakkam/fdtd-2d).  The GPU box never sees the reference, so the vectors
are committed as small .npz files; this script is committed so that they can be
regenerated and audited.  Nothing from the reference's source is copied: the
module is imported and called, and only inputs/outputs are stored.

The reference module deletes and recreates ./frames in the current directory
on import (python-src/main.py:7-9), so it is imported from a scratch cwd.

Each case stores its inputs, the per-step source amplitudes the reference
produced (so a checker can feed the very same float64 numbers and does not
depend on the local libm/NumPy exp), and the reference outputs for
  * float64 arrays  -- what the reference computes by default, and
  * float32 arrays  -- the same reference code handed arrays cast to float32
    (the like-for-like comparison for the fp32 device path).

Usage:  python tests/golden/make_golden.py [g7]     (from the repo root; `g7` writes only that case)
"""
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference/python-src"
OUT = os.path.dirname(os.path.abspath(__file__))

DT = 5e-14      # fdtd.py:16
DX = 1e-4       # fdtd.py:17
FC = 30e9       # fdtd.py:34


def load_reference():
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present; golden vectors are regenerated "
                 "only in the build container")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    scratch = tempfile.mkdtemp(prefix="fdtd_ref_")
    os.chdir(scratch)
    sys.path.insert(0, REF)
    import main as ref  # noqa: E402  (the reference's python-src/main.py)
    return ref


def run_loop(ref, Ez, Hx, Hy, eps, mu, nsteps, src, snaps=()):
    """fdtd.py:30-34 around the imported functions. Returns amps and snapshots."""
    rows, cols = Ez.shape
    amps = np.zeros(nsteps)
    shots = {}
    for i in range(nsteps):
        Hx, Hy = ref.update_Hx_Hy(Ez, Hx, Hy, mu, eps, DT, DX)
        Ez = ref.update_Ez(Ez, Hx, Hy, mu, eps, DT, DX)
        s = ref.ricker(rows, cols, src[0], src[1], i * DT, FC)
        amps[i] = s[src[0], src[1]]
        Ez += s
        if (i + 1) in snaps:
            shots[i + 1] = (Ez.copy(), Hx.copy(), Hy.copy())
    return Ez, Hx, Hy, amps, shots


def single_call_case(ref, name, rows, cols, seed):
    """G1 / G6: update_Hx_Hy and update_Ez in isolation on random state."""
    rng = np.random.default_rng(seed)
    Ez = rng.standard_normal((rows, cols))
    Hx = rng.standard_normal((rows, cols - 1)) * 1e-3
    Hy = rng.standard_normal((rows - 1, cols)) * 1e-3
    eps = 8.85418e-12 * rng.uniform(1.0, 10.0, (rows, cols))
    mu = np.full((rows, cols), 4 * np.pi * 1e-7)
    out = dict(Ez=Ez, Hx=Hx, Hy=Hy, eps=eps, mu=mu, dt=DT, dx=DX)
    for tag, dt_ in (("f64", np.float64), ("f32", np.float32)):
        e, hx, hy = Ez.astype(dt_), Hx.astype(dt_), Hy.astype(dt_)
        ep, m = eps.astype(dt_), mu.astype(dt_)
        hx1, hy1 = ref.update_Hx_Hy(e, hx, hy, m, ep, DT, DX)
        out[f"h_Hx_{tag}"], out[f"h_Hy_{tag}"] = hx1.copy(), hy1.copy()
        # E half-step from the ORIGINAL H (isolates update_Ez) ...
        e2 = ref.update_Ez(Ez.astype(dt_), Hx.astype(dt_), Hy.astype(dt_), m, ep, DT, DX)
        out[f"e_Ez_{tag}"] = e2.copy()
        # ... and one full H->E step
        e3 = ref.update_Ez(e, hx1, hy1, m, ep, DT, DX)
        out[f"step_Ez_{tag}"] = e3.copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def loop_case(ref, name, rows, cols, eps, src, nsteps, snaps, keep_f32_snaps=True):
    out = dict(rows=rows, cols=cols, dt=DT, dx=DX, fc=FC, nsteps=nsteps,
               src=np.array(src), snaps=np.array(sorted(snaps)))
    mu = np.full((rows, cols), 4 * np.pi * 1e-7)
    uniform = bool(np.all(eps == eps.flat[0]))
    if uniform:
        out["eps_uniform"] = eps.flat[0]
    else:
        out["eps"] = eps
    for tag, dt_ in (("f64", np.float64), ("f32", np.float32)):
        Ez, Hx, Hy = ref.grid_init(rows, cols)
        Ez, Hx, Hy = Ez.astype(dt_), Hx.astype(dt_), Hy.astype(dt_)
        Ez, Hx, Hy, amps, shots = run_loop(ref, Ez, Hx, Hy, eps.astype(dt_),
                                           mu.astype(dt_), nsteps, src, snaps)
        assert Ez.dtype == dt_ and Hx.dtype == dt_
        if tag == "f64":
            out["amps"] = amps
        for n, (e, hx, hy) in shots.items():
            if n == nsteps or tag == "f64" or keep_f32_snaps:
                out[f"Ez_{tag}_{n}"], out[f"Hx_{tag}_{n}"], out[f"Hy_{tag}_{n}"] = e, hx, hy
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def g7_case(ref):
    # G7: 2000 steps (the length SURVEY.md M3 states its fp32-vs-fp64 tolerance for): 96x96 vacuum,
    # the pulse crosses the Mur frame several times; final fields only
    r, c = 96, 96
    e, m = ref.material_init(None, r, c)
    loop_case(ref, "g7_vacuum_96x96_2000", r, c, e, (48, 40), 2000, {2000}, keep_f32_snaps=False)


def main():
    ref = load_reference()
    eps0 = 8.85418e-12
    if len(sys.argv) > 1 and sys.argv[1] == "g7":      # add the newest case without rewriting the others
        g7_case(ref)
        print("g7 written")
        return
    g7_case(ref)

    # G1, G6: isolated half-steps (non-square, varying eps; minimum sizes)
    single_call_case(ref, "g1_single_48x40", 48, 40, 1234)
    single_call_case(ref, "g6_single_11x11", 11, 11, 11)
    single_call_case(ref, "g6_single_12x13", 12, 13, 12)

    # G2: vacuum, wave reaches the Mur band (~180 steps at Courant 0.15)
    r, c = 64, 64
    e, m = ref.material_init(None, r, c)
    assert np.all(m == 4 * np.pi * 1e-7)
    loop_case(ref, "g2_vacuum_64x64", r, c, e, (32, 32), 1200, {200, 600, 1200})

    # G3: dielectric disk (eps_r = 10), source near the top-left corner, non-square
    r, c = 64, 80
    ii, jj = np.mgrid[0:r, 0:c]
    eps = np.where((ii - 36) ** 2 + (jj - 44) ** 2 <= 14 ** 2, 10 * eps0, eps0)
    loop_case(ref, "g3_disk_64x80", r, c, eps, (8, 9), 800, {100, 400, 800})

    # G4: BASELINE config 1 -- 256x256 vacuum, 500 steps
    r, c = 256, 256
    e, m = ref.material_init(None, r, c)
    loop_case(ref, "g4_config1_256x256", r, c, e, (128, 128), 500, {500},
              keep_f32_snaps=False)

    # G5: source waveforms and scalar constants
    steps = np.array([0, 1, 100, 333, 666, 667, 1000])
    rick = np.array([ref.ricker(3, 3, 1, 1, i * DT, FC)[1, 1] for i in steps])
    sinu = np.array([ref.sinusoidal(3, 3, 1, 1, i * DT, FC)[1, 1] for i in steps])
    e, m = ref.material_init(None, 4, 4)
    gz = ref.grid_init(7, 9)
    np.savez_compressed(
        os.path.join(OUT, "g5_scalars.npz"),
        steps=steps, ricker=rick, sinusoidal=sinu, dt=DT, fc=FC,
        eps_vac=e[0, 0], mu_vac=m[0, 0],
        grid_shapes=np.array([a.shape for a in gz]),
        grid_dtype=str(gz[0].dtype))
    # N2: material_init(path, ...) -- grayscale structure image -> eps (main.py:108-123).
    # The image is drawn here (a bus waveguide and a ring, white background, black = core);
    # the reference's own asset python-src/assets/example_structure.png is not in its repo.
    from PIL import Image, ImageDraw
    img = Image.new("L", (120, 96), 255)
    d = ImageDraw.Draw(img)
    d.line([(0, 18), (119, 18)], fill=0, width=5)
    d.ellipse([30, 30, 90, 90], outline=0, width=4)
    d.rectangle([100, 60, 110, 80], fill=128)
    png = os.path.join(OUT, "structure_120x96.png")
    img.save(png)
    for (rr, cc, bp) in ((64, 72, 10.0), (96, 120, 4.0)):
        e, m = ref.material_init(png, rr, cc, bp)
        np.savez_compressed(os.path.join(OUT, f"n2_material_{rr}x{cc}.npz"), eps=e, mu=m,
                            rows=rr, cols=cc, black_point=bp)

    # N1: capture_snapshot (main.py:153-179) on a mid-run field with a non-uniform eps map
    g3 = np.load(os.path.join(OUT, "g3_disk_64x80.npz"))
    for tag, Ezs in (("f64", g3["Ez_f64_400"]), ("f32", g3["Ez_f32_400"])):
        path = os.path.join(tempfile.mkdtemp(prefix="fdtd_snap_"), "s.png")
        ref.capture_snapshot(Ezs, g3["eps"], path, 1e-3, -1e-3)
        rgb = np.array(Image.open(path))
        path2 = path.replace("s.png", "u.png")
        ref.capture_snapshot(Ezs, np.full_like(g3["eps"], 8.85418e-12), path2, 0.2, -0.2)
        rgb_u = np.array(Image.open(path2))
        np.savez_compressed(os.path.join(OUT, f"n1_snapshot_{tag}.npz"), rgb=rgb, rgb_uniform=rgb_u,
                            vmax=1e-3, vmin=-1e-3, vmax_u=0.2, vmin_u=-0.2)

    # N2: the reference's structure canvas (region_drawer.py:5-87) -- every primitive once
    import region_drawer as rd
    cv = rd.RegionDrawer(160, 120)
    cv.draw_waveguide((0, 20), (159, 20), 6)
    cv.draw_waveguide((10, 110), (150, 95), 5)
    cv.draw_ring_resonator((70, 66), 30, 5)
    cv.draw_sphere((135, 60), 9, 4)
    cv.draw_curved_waveguide((5, 40), (60, 115), (8, 100), 4)
    cv.draw_directional_coupler((90, 108), 60, 7, 3)
    png = os.path.join(tempfile.mkdtemp(prefix="fdtd_canvas_"), "canvas.png")
    cv.save(png)
    e, m = ref.material_init(png, 90, 128, 6.0)
    np.savez_compressed(os.path.join(OUT, "n2_structure_canvas.npz"), pixels=np.array(Image.open(png)),
                        eps=e, mu=m, rows=90, cols=128, black_point=6.0)

    print("golden vectors written to", OUT)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
