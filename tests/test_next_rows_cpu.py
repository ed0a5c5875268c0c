"""CPU: the callers either side of the hot path against vectors
produced by running the reference: N2 material_init(path) and N1 capture_snapshot."""
import os

import numpy as np
import pytest


@pytest.mark.parametrize("name", ["n2_material_64x72", "n2_material_96x120"])
def test_material_init_from_image_matches_reference(golden_dir, name):
    import fdtd2d_amd as fd
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    eps, mu = fd.material_init(os.path.join(golden_dir, "structure_120x96.png"), int(g["rows"]),
                               int(g["cols"]), float(g["black_point"]))
    assert eps.dtype == np.float64 and np.array_equal(eps, g["eps"]) and np.array_equal(mu, g["mu"])
    assert eps.max() > 2 * fd.EPS0 and eps.min() == pytest.approx(fd.EPS0)


@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
def test_capture_snapshot_matches_reference_pixels(golden_dir, tmp_path, tag, dtype):
    import fdtd2d_amd as fd
    from PIL import Image
    g3 = np.load(os.path.join(golden_dir, "g3_disk_64x80.npz"))
    g = np.load(os.path.join(golden_dir, f"n1_snapshot_{tag}.npz"))
    Ez = g3[f"Ez_{tag}_400"]
    assert Ez.dtype == dtype
    path = os.path.join(str(tmp_path), "s.png")
    img = fd.capture_snapshot(Ez, g3["eps"], path, float(g["vmax"]), float(g["vmin"]))
    assert np.array_equal(img, g["rgb"]) and np.array_equal(np.array(Image.open(path)), g["rgb"])
    uni = fd.capture_snapshot(Ez, np.full_like(g3["eps"], fd.EPS0), None, float(g["vmax_u"]), float(g["vmin_u"]))
    assert np.array_equal(uni, g["rgb_uniform"])
    assert len(np.unique(g["rgb"].reshape(-1, 3), axis=0)) > 20      # a real picture, not a flat one
