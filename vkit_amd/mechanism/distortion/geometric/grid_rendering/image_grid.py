"""``ImageGrid``: the vertex lattice of an image-grid distortion (reference: grid_rendering/type.py).

Array-backed instead of lists of ``Point`` objects: ``smooth`` holds the float vertex positions as (x, y) and
``vertices`` their rounded integer pixel positions (int32, Python ``round`` = half-to-even), which is all the
downstream remap consumes.  The dense remap of the reference
(``generate_remap_params``, type.py:209-261) is not materialised on the host: ``grid_remap`` hands both integer
lattices to the HIP kernels, ``generate_remap_params`` exists for callers that want the float32 maps.
"""
from typing import Dict, Optional, Tuple

import numpy as np

from vkit_amd import _native
from vkit_amd.element import Point, PointList, Polygon
from .homography import get_perspective_transform


class ImageGrid:

    def __init__(self, smooth: np.ndarray, grid_size: Optional[int] = None):
        smooth = np.asarray(smooth, dtype=np.float64)
        assert smooth.ndim == 3 and smooth.shape[2] == 2
        self.smooth = smooth
        self.vertices = np.ascontiguousarray(np.rint(smooth).astype(np.int32))
        # set for a source grid only
        self.grid_size = grid_size
        self._trans_mats: Dict[Tuple[int, int, bool], np.ndarray] = {}
        self._maps = None
        assert self.vertices[..., 1].min() == 0 and self.vertices[..., 0].min() == 0
        self.image_height = int(self.vertices[..., 1].max()) + 1
        self.image_width = int(self.vertices[..., 0].max()) + 1

    # ---- shape
    @property
    def image_shape(self):
        return self.image_height, self.image_width

    @property
    def num_rows(self):
        return self.smooth.shape[0]

    @property
    def num_cols(self):
        return self.smooth.shape[1]

    @property
    def shape(self):
        return self.num_rows, self.num_cols

    def compatible_with(self, other: 'ImageGrid'):
        return self.shape == other.shape

    # ---- Point views (API compatibility / debugging)
    def point(self, row: int, col: int) -> Point:
        x, y = self.smooth[row, col]
        return Point.create(y=float(y), x=float(x))

    @property
    def points_2d(self):
        return [PointList(self.point(r, c) for c in range(self.num_cols)) for r in range(self.num_rows)]

    @property
    def flatten_points(self):
        return PointList(self.point(r, c) for r in range(self.num_rows) for c in range(self.num_cols))

    def generate_polygon(self, polygon_row: int, polygon_col: int):
        r, c = polygon_row, polygon_col
        # clockwise from the cell's (row, col) vertex
        return Polygon.create(points=(self.point(r, c), self.point(r, c + 1), self.point(r + 1, c + 1),
                                      self.point(r + 1, c)))

    def generate_border_polygon(self):
        """Clockwise outline of the lattice (reference type.py:128-141)."""
        rows, cols = self.shape
        ring = [(0, c) for c in range(cols)]
        ring += [(r, cols - 1) for r in range(1, rows)]
        ring += [(rows - 1, c) for c in reversed(range(cols - 1))]
        ring += [(r, 0) for r in reversed(range(1, rows - 1))]
        return Polygon.create(points=PointList(self.point(r, c) for r, c in ring))

    def _cell_quad(self, row: int, col: int) -> np.ndarray:
        v = self.vertices
        return np.asarray([v[row, col], v[row, col + 1], v[row + 1, col + 1], v[row + 1, col]], dtype=np.float32)

    def get_trans_mat(self, polygon_row: int, polygon_col: int, other: 'ImageGrid'):
        """Homography of one cell, self -> other (cached), from the ROUNDED quads (reference type.py:166-180)."""
        key = (polygon_row, polygon_col, False)
        if key not in self._trans_mats:
            self._trans_mats[key] = get_perspective_transform(self._cell_quad(polygon_row, polygon_col),
                                                              other._cell_quad(polygon_row, polygon_col))
        return self._trans_mats[key]

    def get_inv_trans_mat(self, polygon_row: int, polygon_col: int, other: 'ImageGrid'):
        key = (polygon_row, polygon_col, True)
        if key not in self._trans_mats:
            self._trans_mats[key] = get_perspective_transform(other._cell_quad(polygon_row, polygon_col),
                                                              self._cell_quad(polygon_row, polygon_col))
        return self._trans_mats[key]

    def generate_remap_params(self, dst_image_grid: 'ImageGrid'):
        """(map_y, map_x) float32 of the destination shape, computed on the GPU (cached on the source grid)."""
        if self._maps is None:
            map_x, map_y = _native.grid_to_map(self.vertices, dst_image_grid.vertices, dst_image_grid.image_shape)
            self._maps = (map_y, map_x)
        return self._maps

    def to_conducted_resized_image_grid(self, shapable_or_shape, resized_height: int, resized_width: int):
        from vkit_amd.element.opt import extract_shape_from_shapable_or_shape
        height, width = extract_shape_from_shapable_or_shape(shapable_or_shape)
        out = np.empty_like(self.smooth)
        out[..., 1] = np.clip(self.smooth[..., 1] * resized_height / height, 0, resized_height - 1)
        out[..., 0] = np.clip(self.smooth[..., 0] * resized_width / width, 0, resized_width - 1)
        return ImageGrid(out)
