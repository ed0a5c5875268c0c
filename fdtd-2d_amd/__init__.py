"""fdtd-2d_amd: MI355X-native engine for the 2D TE-mode FDTD leapfrog of
skunnavakkam/fdtd-2d (python-src/main.py update_Hx_Hy / update_Ez / ricker as looped by
python-src/fdtd.py).  Import it as ``fdtd2d_amd`` (a directory name with '-' is not an
identifier; ``fdtd2d_amd/__init__.py`` at the repo root points here)."""
from ._abi import ARITHMETIC, Fdtd2dError, LIB_PATH  # noqa: F401
from .api import (EPS0, MU0, capture_snapshot, courant_number, grid_init, invalidate_cache, material_init,  # noqa: F401
                  pml_profiles, render_snapshot, ricker, ricker_amplitude, run_fdtd, sinusoidal,
                  sinusoidal_amplitude, snapshot_indices, eps_background, step, update_Ez,
                  update_Hx_Hy)
from .engine import Engine  # noqa: F401
from .structure import Structure, ring_resonator  # noqa: F401

__all__ = ["Engine", "Structure", "ring_resonator", "Fdtd2dError", "EPS0", "MU0", "grid_init", "material_init", "ricker",
           "sinusoidal", "ricker_amplitude", "sinusoidal_amplitude", "courant_number",
           "update_Hx_Hy", "update_Ez", "step", "run_fdtd", "capture_snapshot", "render_snapshot",
           "snapshot_indices", "eps_background", "pml_profiles", "invalidate_cache"]
