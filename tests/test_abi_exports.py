"""CPU-only: the C-ABI library loads and exports every symbol include/fdtd2d.h declares;
without a GPU it fails loudly instead of falling back to anything."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "fdtd2d.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fdtd2d_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound():
    from fdtd2d_amd import _abi
    names = _declared()
    assert len(names) >= 25
    lib = ctypes.CDLL(_abi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in fdtd2d.h but not exported"
    assert sorted(_abi.SIGNATURES) == names, "ctypes table and header disagree"


def test_header_constants_match_binding():
    from fdtd2d_amd import _abi
    txt = open(os.path.join(ROOT, "include", "fdtd2d.h")).read()
    defs = {k: int(v.strip("()")) for k, v in re.findall(r"#define\s+FDTD2D_(\w+)\s+(\(?-?\d+\)?)", txt)}
    for k, v in defs.items():
        if hasattr(_abi, k):
            assert getattr(_abi, k) == v, k
    assert defs["F32"] == 0 and defs["BOUNDARY_MUR5"] == 1 and defs["E_NODEVICE"] == -2


def test_package_mirrors_reference_surface():
    import inspect
    import fdtd2d_amd as fd
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(fd.update_Hx_Hy) == ["Ez", "Hx", "Hy", "mu", "eps", "dt", "dx"]   # main.py:66
    assert sig(fd.update_Ez) == ["Ez", "Hx", "Hy", "mu", "eps", "dt", "dx"]      # main.py:12
    assert sig(fd.ricker) == ["rows", "cols", "x_pos", "y_pos", "t", "fc"]      # main.py:182
    assert sig(fd.sinusoidal) == ["rows", "cols", "x_pos", "y_pos", "t", "fc"]  # main.py:190
    assert sig(fd.material_init) == ["path", "rows", "cols", "black_point"]     # main.py:88
    assert sig(fd.step)[:7] == ["E", "Hx", "Hy", "eps", "mu", "source", "t"]     # north_star
    Ez, Hx, Hy = fd.grid_init(7, 9)
    assert (Ez.shape, Hx.shape, Hy.shape) == ((7, 9), (7, 8), (6, 9)) and Ez.dtype == np.float64
    eps, mu = fd.material_init(None, 3, 4)
    assert eps[0, 0] == 8.85418e-12 and mu[0, 0] == 4 * np.pi * 1e-7


def test_host_waveforms_match_golden(golden_dir):
    import fdtd2d_amd as fd
    from fdtd2d_amd import _abi
    g = np.load(os.path.join(golden_dir, "g5_scalars.npz"))
    dt, fc = float(g["dt"]), float(g["fc"])
    lib = _abi.load()
    for i, r, s in zip(g["steps"], g["ricker"], g["sinusoidal"]):
        t = int(i) * dt
        assert fd.ricker(5, 6, 2, 3, t, fc)[2, 3] == pytest.approx(r, rel=4e-16)
        assert fd.sinusoidal(5, 6, 2, 3, t, fc)[2, 3] == pytest.approx(s, rel=4e-16, abs=1e-300)
        assert lib.fdtd2d_source_amplitude(_abi.SRC_RICKER, t, fc) == pytest.approx(r, rel=1e-14)
        assert lib.fdtd2d_source_amplitude(_abi.SRC_SINUSOIDAL, t, fc) == pytest.approx(s, rel=1e-13, abs=1e-300)


def test_no_cpu_fallback_without_a_device():
    """In the build container there is no GPU: creation must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import fdtd2d_amd as fd
    with pytest.raises(fd.Fdtd2dError) as ei:
        fd.Engine(64, 64)
    assert ei.value.code == -2 and "no CPU path" in str(ei.value)
    with pytest.raises(fd.Fdtd2dError):
        Ez, Hx, Hy = fd.grid_init(16, 16)
        eps, mu = fd.material_init(None, 16, 16)
        fd.update_Hx_Hy(Ez, Hx, Hy, mu, eps, 5e-14, 1e-4)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fdtd-2d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "fdtd_oracle" not in src and "libfdtd_oracle" not in src, f
