"""Source lattice creation and its projection to the destination lattice (reference:
grid_rendering/grid_creator.py:22-115): vertices every ``grid_size`` pixels plus the last row / column; the
projected lattice is shifted so that its rounded minimum sits at the origin and optionally rescaled to the
source size."""
import numpy as np

from .image_grid import ImageGrid
from .point_projector import PointProjector


def _axis(length: int, grid_size: int):
    ticks = list(range(0, length, grid_size))
    if ticks[-1] != length - 1:
        ticks.append(length - 1)
    return ticks


def create_src_image_grid(height: int, width: int, grid_size: int):
    ys, xs = _axis(height, grid_size), _axis(width, grid_size)
    smooth = np.empty((len(ys), len(xs), 2), dtype=np.float64)
    smooth[..., 0] = np.asarray(xs, dtype=np.float64)[None, :]
    smooth[..., 1] = np.asarray(ys, dtype=np.float64)[:, None]
    return ImageGrid(smooth, grid_size=grid_size)


def create_dst_image_grid_and_shift_amounts_and_resize_ratios(src_image_grid: ImageGrid,
                                                              point_projector: PointProjector,
                                                              resize_as_src: bool = True):
    rows, cols = src_image_grid.shape
    # the lattice stays an array from end to end (10 816 vertices at 2048^2: no per-vertex Point objects)
    projected = np.asarray(point_projector.project_array(src_image_grid.smooth.reshape(-1, 2)), dtype=np.float64)
    assert projected.shape == (rows * cols, 2)
    if not np.isfinite(projected).all():
        # the reference builds a Point per vertex and fails in its round() (element/point.py:31-47)
        if np.isnan(projected).any():
            raise ValueError('cannot convert float NaN to integer')
        raise OverflowError('cannot convert float infinity to integer')
    smooth = np.array(projected.reshape(rows, cols, 2))

    # the shift is the minimum of the ROUNDED positions (an integer), applied to the smooth positions
    rounded = np.rint(smooth)
    shift_amount_y = int(rounded[..., 1].min())
    shift_amount_x = int(rounded[..., 0].min())
    smooth[..., 1] = smooth[..., 1] + (-shift_amount_y)
    smooth[..., 0] = smooth[..., 0] + (-shift_amount_x)

    resize_ratio_y = resize_ratio_x = 1.0
    if resize_as_src:
        raw = ImageGrid(smooth)
        src_h, src_w = src_image_grid.image_shape
        resize_ratio_y = src_h / raw.image_height
        resize_ratio_x = src_w / raw.image_width
        dst_image_grid = raw.to_conducted_resized_image_grid(raw.image_shape, resized_height=src_h,
                                                             resized_width=src_w)
        assert dst_image_grid.image_shape == (src_h, src_w)
    else:
        dst_image_grid = ImageGrid(smooth)
    return dst_image_grid, (shift_amount_y, shift_amount_x), (resize_ratio_y, resize_ratio_x)


def create_dst_image_grid(src_image_grid: ImageGrid, point_projector: PointProjector, resize_as_src: bool = True):
    return create_dst_image_grid_and_shift_amounts_and_resize_ratios(src_image_grid, point_projector,
                                                                     resize_as_src=resize_as_src)[0]
