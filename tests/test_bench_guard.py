"""bench.py's cut-band check (the correctness guard of `bench.py --gpus N`): it must accept rows that equal the
single-domain result and reject rows in which a halo arrived wrong.  One GPU, no process group: a stand-in runner
object describes "rank 0 of 2" whose owned rows come from a whole-grid run."""
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_hash_rows_is_a_function_of_the_global_index():
    import bench
    a = bench.hash_rows(100, 140, 300, 1, 1.0)
    b = bench.hash_rows(0, 200, 300, 1, 1.0)
    assert a.dtype == np.float32 and np.array_equal(a, b[100:140]) and np.abs(a).max() <= 1 and a.std() > 0.3
    assert not np.array_equal(a, bench.hash_rows(100, 140, 300, 2, 1.0))


@pytest.mark.gpu
@pytest.mark.parametrize("boundary,init", [("mur", "hash"), ("mur", "zero"), ("pml", "hash")])
def test_cut_band_check_accepts_the_truth_and_rejects_a_wrong_halo(boundary, init):
    import bench
    import fdtd2d_amd as fd
    rows, cols, steps, cut = 700, 900, 40, 350
    src = (cut, cols // 2)
    amps = bench.amplitudes(fd, 380, steps)
    with fd.Engine(rows, cols, bench.DT, bench.DX, dtype=np.float32, boundary=boundary) as eng:
        eng.set_materials()
        if boundary == "pml":
            eng.set_pml()
        if init == "hash":
            eng.upload(bench.hash_rows(0, rows, cols, 1, 1.0), bench.hash_rows(0, rows, cols, 2, 1e-3)[:, :cols - 1],
                       bench.hash_rows(0, rows - 1, cols, 3, 1e-3))
        eng.run(steps, src[0], src[1], amps)
        full = eng.download()
    for rank, (r0, r1) in enumerate(((0, cut), (cut, rows))):
        runner = types.SimpleNamespace(r0=r0, r1=r1, rank=rank, up=None if rank == 0 else 0, down=1 if rank == 0 else None)
        owned = [full[0][r0:r1].copy(), full[1][r0:r1].copy(), full[2][r0:min(r1, rows - 1)].copy()]
        ok, n = bench.check_cut_bands(fd, runner, rows, cols, "uniform", boundary, 0, owned, steps, src, amps, init)
        assert ok and n == 3 * 24
        # a halo that arrived one row off shows up as wrong values in the rows next to the cut
        row = (r1 - 3 - r0) if rank == 0 else 2
        owned[0][row, 400:410] += np.float32(1e-3)
        ok, _ = bench.check_cut_bands(fd, runner, rows, cols, "uniform", boundary, 0, owned, steps, src, amps, init)
        assert not ok
