"""The tolerance build, libfdtd2d_fused.so (FDTD2D_ARITHMETIC=fused): the same sources compiled with a*b+c
contracted into FMA -- 8 instead of 11 operations per cell-step ("keep -ffp-contract=off for
the parity build variant, measure the perf difference"; north_star: "within a stated fp32 tolerance").

Stated tolerances (e = max|x - ref| / max|ref| per field):
  float32 fused vs the float64 reference                 e <= 5e-6 at config 1's 500 steps, <= 1e-4 at 2000 steps -- the
                                                         bars the value-identical float32 build is held to (it measures
                                                         4.9e-5 at 2000 steps, the fused build 3.8e-5: one rounding less
                                                         per multiply-add)
  float32 fused vs the reference run on float32 arrays   e <= 1e-5 up to 800 steps, <= 2e-4 at 2000 steps (two float32
                                                         runs that round differently drift apart like each drifts from
                                                         float64: measured 7.1e-5)
  float64 fused vs the float64 reference                 e <= 1e-12
and the fused build is still deterministic in the launch shape: temporally blocked passes equal its own single-step
kernels bit for bit (every kernel contracts the same expressions the same way).

The library is chosen when the package is first imported, so the GPU part runs in a child process."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, ROOT)
import fdtd2d_amd as fd
from fdtd2d_amd import _abi
assert fd.ARITHMETIC == "fused" and _abi.LIB_PATH.endswith("libfdtd2d_fused.so")
ver = _abi.load().fdtd2d_version().decode()
DT, DX = 5e-14, 1e-4
out = {"version": ver}
rel = lambda a, ref: float(np.abs(a.astype(np.float64) - ref).max() / np.abs(ref).max())
G = os.path.join(ROOT, "tests", "golden")
# golden G7: 96 x 96 vacuum, 2000 steps of the reference
g = np.load(os.path.join(G, "g7_vacuum_96x96_2000.npz"))
sr, sc = (int(v) for v in g["src"])
for tag, dtype in (("f32", np.float32), ("f64", np.float64)):
    with fd.Engine(96, 96, DT, DX, dtype=dtype) as eng:
        eng.set_materials()
        eng.run(2000, sr, sc, g["amps"])
        got = eng.download()
    out["g7_" + tag + "_vs_same_type"] = max(rel(a, g[f"{k}_{tag}_2000"]) for a, k in zip(got, ("Ez", "Hx", "Hy")))
    out["g7_" + tag + "_vs_f64"] = max(rel(a, g[f"{k}_f64_2000"]) for a, k in zip(got, ("Ez", "Hx", "Hy")))
# golden G4 = BASELINE configs[0]: 256 x 256 vacuum, 500 steps
g = np.load(os.path.join(G, "g4_config1_256x256.npz"))
got = fd.run_fdtd(256, 256, DT, DX, 500, dtype=np.float32)
out["g4_f32_vs_f32"] = max(rel(a, g[f"{k}_f32_500"]) for a, k in zip(got, ("Ez", "Hx", "Hy")))
out["g4_f32_vs_f64"] = max(rel(a, g[f"{k}_f64_500"]) for a, k in zip(got, ("Ez", "Hx", "Hy")))
# golden G3: dielectric disk (array eps), 800 steps
g = np.load(os.path.join(G, "g3_disk_64x80.npz"))
r, c = int(g["rows"]), int(g["cols"])
s_last = int(g["snaps"][-1])
with fd.Engine(r, c, DT, DX, dtype=np.float32) as eng:
    eng.set_materials(g["eps"].astype(np.float32), np.full((r, c), fd.MU0, np.float32))
    eng.run(s_last, int(g["src"][0]), int(g["src"][1]), g["amps"])
    got = eng.download()
out["g3_f32_vs_f32"] = max(rel(a, g[f"{k}_f32_{s_last}"]) for a, k in zip(got, ("Ez", "Hx", "Hy")))
# launch-shape independence inside the fused build: passes (16 + 16 + 20-step kernels, zones, edge strips, array
# eps) == single-step kernels, bit for bit, from a random state
rng = np.random.default_rng(3)
R, C, n = 300, 1100, 52
st = [rng.standard_normal((R, C), dtype=np.float32), rng.standard_normal((R, C - 1), dtype=np.float32) * np.float32(1e-3),
      rng.standard_normal((R - 1, C), dtype=np.float32) * np.float32(1e-3)]
eps = (fd.EPS0 * rng.uniform(1, 10, (R, C))).astype(np.float32)
amps = rng.standard_normal(n)
res = []
for max_nt in (20, 16, 8, 0):
    with fd.Engine(R, C, DT, DX, dtype=np.float32) as eng:
        eng.set_materials(eps, np.float32(fd.MU0)).set_option(max_pass_steps=max_nt)
        eng.upload(*st)
        eng.run(n, 21, 223, amps)
        res.append(eng.download())
out["passes_equal_steps"] = [bool(all(np.array_equal(a, b) for a, b in zip(res[k], res[3]))) for k in range(3)]
out["passes_vs_steps_rel"] = [max(rel(a, b.astype(np.float64)) for a, b in zip(res[k], res[3])) for k in range(3)]
print("FUSED_RESULT " + json.dumps(out))
'''


def test_both_builds_export_the_whole_abi():
    """CPU: libfdtd2d_fused.so exists next to libfdtd2d.so and exports every symbol the header declares."""
    import ctypes
    import re
    txt = open(os.path.join(ROOT, "include", "fdtd2d.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = sorted(set(re.findall(r"\b(fdtd2d_[a-z0-9_]+)\s*\(", txt)))
    for lib in ("libfdtd2d.so", "libfdtd2d_fused.so"):
        h = ctypes.CDLL(os.path.join(ROOT, "fdtd-2d_amd", lib))
        for n in names:
            assert hasattr(h, n), f"{lib}: {n} not exported"
        h.fdtd2d_version.restype = ctypes.c_char_p
        v = h.fdtd2d_version().decode()
        assert ("fused" in v) == ("fused" in lib), v


def test_arithmetic_selection_is_validated():
    p = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); import fdtd2d_amd"],
                       capture_output=True, text=True, env=dict(os.environ, FDTD2D_ARITHMETIC="sloppy"))
    assert p.returncode != 0 and "FDTD2D_ARITHMETIC" in p.stderr


@pytest.mark.gpu
def test_fused_build_within_the_stated_tolerance():
    p = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, FDTD2D_ARITHMETIC="fused"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("FUSED_RESULT ")][-1][13:])
    assert "fused" in out["version"]
    print(out)
    assert out["g7_f32_vs_same_type"] <= 2e-4 and out["g7_f32_vs_f64"] <= 1e-4, out
    assert out["g7_f64_vs_f64"] <= 1e-12, out
    assert out["g4_f32_vs_f32"] <= 1e-5 and out["g4_f32_vs_f64"] <= 5e-6, out
    assert out["g3_f32_vs_f32"] <= 1e-5, out
    assert all(out["passes_equal_steps"]), out
