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


def _golden_canvas():
    from fdtd2d_amd.structure import Structure
    s = Structure(160, 120)
    s.waveguide((0, 20), (159, 20), 6).waveguide((10, 110), (150, 95), 5)
    s.ring((70, 66), 30, 5).disk((135, 60), 9, 4)
    s.bend((5, 40), (60, 115), (8, 100), 4).coupler((90, 108), 60, 7, 3)
    return s


def test_structure_canvas_matches_reference_drawer(golden_dir, tmp_path):
    """N2: every primitive of the structure canvas rasterises like the reference's
    region_drawer.py (pixels of a canvas drawn BY the reference, make_golden.py), and
    materials() equals the reference's material_init on the saved image."""
    import fdtd2d_amd as fd
    g = np.load(os.path.join(golden_dir, "n2_structure_canvas.npz"))
    s = _golden_canvas()
    assert np.array_equal(s.pixels(), g["pixels"])
    rows, cols, bp = int(g["rows"]), int(g["cols"]), float(g["black_point"])
    eps, mu = s.materials(rows, cols, bp)
    assert np.array_equal(eps, g["eps"]) and np.array_equal(mu, g["mu"])
    path = os.path.join(str(tmp_path), "canvas.png")
    s.save(path)
    eps2, mu2 = fd.material_init(path, rows, cols, bp)
    assert np.array_equal(eps2, g["eps"]) and np.array_equal(mu2, g["mu"])


def test_ring_resonator_recipe_shape():
    import fdtd2d_amd as fd
    eps, mu = fd.ring_resonator(200, 240)
    assert eps.shape == mu.shape == (200, 240)
    assert eps.min() == fd.EPS0 and np.isclose(eps.max(), 10 * fd.EPS0)
    assert eps[40, 120] == eps.max() and eps[108, 120] == fd.EPS0        # bus core, ring centre
    assert np.all(mu == fd.MU0)
