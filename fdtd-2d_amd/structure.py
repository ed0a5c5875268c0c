"""Structure images for `material_init(path, ...)` -- SURVEY.md section 8(f) row N2.

The reference draws its permittivity maps as 8-bit grayscale images (white = background,
black = core) with a small PIL canvas (python-src/region_drawer.py:5-87) and turns them into
eps with material_init (main.py:108-123).  `Structure` is that canvas under this package's
names; the primitives rasterise exactly like the reference's (checked pixel for pixel against
a canvas drawn by the reference, tests/golden/n2_structure_canvas.npz), and `materials()`
goes straight to (eps, mu) without a file in between.
"""
from __future__ import annotations

import numpy as np

from .api import EPS0, MU0


class Structure:
    CORE, BACKGROUND = 0, 255

    def __init__(self, width: int, height: int):
        from PIL import Image, ImageDraw
        self.width, self.height = int(width), int(height)
        self.image = Image.new("L", (self.width, self.height), self.BACKGROUND)
        self._pen = ImageDraw.Draw(self.image)

    # -- primitives (region_drawer.py:13-83) ------------------------------------------------
    def waveguide(self, start, end, width: int):
        """Straight core strip from `start` to `end` (x, y), `width` pixels wide."""
        self._pen.line([tuple(start), tuple(end)], fill=self.CORE, width=int(width))
        return self

    def _bbox(self, center, radius, width):
        half = radius + width // 2
        return [center[0] - half, center[1] - half, center[0] + half, center[1] + half]

    def ring(self, center, radius: int, width: int):
        """Ring resonator: circle outline of mean radius `radius`, `width` pixels wide."""
        self._pen.ellipse(self._bbox(center, radius, width), outline=self.CORE, width=int(width))
        return self

    def disk(self, center, radius: int, rim: int = 0):
        """Filled disk (the reference's draw_sphere: radius + rim // 2 pixels)."""
        self._pen.ellipse(self._bbox(center, radius, rim), fill=self.CORE)
        return self

    def bend(self, start, end, control, width: int, samples: int = 100):
        """Curved waveguide along the quadratic Bezier curve start -> (control) -> end."""
        t = np.linspace(0, 1, samples)
        pts = [((1 - s) ** 2 * start[0] + 2 * (1 - s) * s * control[0] + s ** 2 * end[0],
                (1 - s) ** 2 * start[1] + 2 * (1 - s) * s * control[1] + s ** 2 * end[1]) for s in t]
        self._pen.line(pts, fill=self.CORE, width=int(width))
        return self

    def coupler(self, start, length: int, gap: int, width: int):
        """Directional coupler: two parallel waveguides `gap` apart, centred on `start`'s y."""
        off = gap // 2 + width // 2
        for dy in (-off, off):
            self.waveguide((start[0], start[1] + dy), (start[0] + length, start[1] + dy), width)
        return self

    # -- outputs -----------------------------------------------------------------------------
    def pixels(self) -> np.ndarray:
        return np.array(self.image, dtype=np.uint8)

    def save(self, path: str):
        self.image.save(path)
        return self

    def materials(self, rows: int, cols: int, black_point: float = 10.0):
        """(eps, mu) as material_init(path, rows, cols, black_point) would return them for this
        canvas saved as a PNG (main.py:108-123): LANCZOS resize to (cols, rows), linear map
        white -> EPS0, black -> black_point * EPS0."""
        from PIL import Image
        img = self.image.resize((int(cols), int(rows)), Image.LANCZOS)
        darkness = 1.0 - np.array(img, dtype=float) / 255.0
        return (1 + (black_point - 1) * darkness) * EPS0, np.ones((rows, cols)) * MU0


def ring_resonator(rows: int, cols: int, black_point: float = 10.0):
    """Bus waveguide + ring in the proportions of BASELINE configs[2],
    drawn on a canvas of the grid's own size: (eps, mu)."""
    s = Structure(cols, rows)
    s.waveguide((0, int(0.20 * rows)), (cols - 1, int(0.20 * rows)), max(1, int(0.04 * rows)))
    s.ring((int(0.50 * cols), int(0.54 * rows)), int(0.30 * rows), max(1, int(0.04 * rows)))
    return s.materials(rows, cols, black_point)
