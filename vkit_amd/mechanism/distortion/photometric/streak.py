"""Streak distortions (reference: photometric/streak.py): periodic line stripes, concentric rectangles and concentric
ellipses alpha-blended onto the image -- vertical structures first, then horizontal ones, so crossings are blended
twice.  ``line_streak`` evaluates its stripe masks analytically inside one HIP kernel; ``rectangle_streak``
builds its two bar masks with index arithmetic on the host and blends them as two composite layers;
``ellipse_streak`` rasterises the ``cv.ellipse`` outlines on the device and blends them as one layer."""
from typing import List, Optional, Tuple

import attrs
import numpy as np
from numpy.random import Generator as RandomGenerator

from vkit_amd import _native
from vkit_amd.element import Box, Image
from ..interface import Distortion, DistortionConfig, DistortionNopState


@attrs.define
class LineStreakConfig(DistortionConfig):
    thickness: int = 1
    gap: int = 4
    dash_thickness: int = 0
    dash_gap: int = 0
    color: Tuple[int, int, int] = (0, 0, 0)
    alpha: float = 1.0
    enable_vert: bool = True
    enable_hori: bool = True


def _check_color(image: Image, color):
    if image.arr.ndim != 3 or len(color) != image.arr.shape[2]:
        raise RuntimeError('value is tuple but len(value) != num_channels.')


def line_streak_image(config: LineStreakConfig, state, image: Image, rng: Optional[RandomGenerator]):
    _check_color(image, config.color)
    if not isinstance(config.alpha, float):
        raise AttributeError('alpha must be a float')
    if config.alpha < 0.0 or config.alpha > 1.0:
        raise RuntimeError(f'alpha={config.alpha} is invalid.')
    mat = _native.line_streak(image.arr, config.thickness, config.gap, config.dash_thickness, config.dash_gap,
                              config.color, config.alpha, config.enable_vert, config.enable_hori)
    return attrs.evolve(image, mat=mat)


line_streak = Distortion(
    config_cls=LineStreakConfig,
    state_cls=DistortionNopState[LineStreakConfig],
    func_image=line_streak_image,
)


def generate_centered_boxes(height: int, width: int, aspect_ratio: float, short_side_min: int, short_side_step: int):
    """Concentric boxes around the image centre, growing until neither side fits any more."""
    center_y, center_x = height // 2, width // 2
    boxes: List[Box] = []
    while True:
        short_side = short_side_min + len(boxes) * short_side_step
        if aspect_ratio >= 1:
            box_h = short_side
            box_w = round(box_h * aspect_ratio)
        elif 0 < aspect_ratio < 1:
            box_w = short_side
            box_h = round(box_w / aspect_ratio)
        else:
            raise NotImplementedError()
        up = center_y - box_h // 2
        down = up + box_h - 1
        left = center_x - box_w // 2
        right = left + box_w - 1
        if not ((0 <= up and down < height) or (0 <= left and right < width)):
            return boxes
        boxes.append(Box(up=up, down=down, left=left, right=right))


@attrs.define
class RectangleStreakConfig(DistortionConfig):
    thickness: int = 1
    aspect_ratio: Optional[float] = None
    dash_thickness: int = 0
    dash_gap: int = 0
    short_side_min: int = 10
    short_side_step: int = 10
    color: Tuple[int, int, int] = (0, 0, 0)
    alpha: float = 1.0


def _rectangle_bar_masks(height: int, width: int, boxes: List[Box], thickness: int):
    vert = np.zeros((height, width), np.uint8)
    hori = np.zeros((height, width), np.uint8)
    for box in boxes:
        inner_up, inner_down = box.down - thickness + 1, box.up + thickness - 1
        inner_left, inner_right = box.right - thickness + 1, box.left + thickness - 1
        rows = slice(max(0, box.up), min(height - 1, box.down) + 1)
        if rows.start < rows.stop:
            if 0 <= inner_right < width:     # left bar
                vert[rows, max(0, box.left):inner_right + 1] = 1
            if 0 <= inner_left < width:      # right bar
                vert[rows, inner_left:min(width - 1, box.right) + 1] = 1
        cols = slice(max(0, inner_right + 1), min(width - 1, inner_left - 1) + 1)
        if cols.start < cols.stop:
            if 0 <= inner_down < height:     # top bar
                hori[max(0, box.up):inner_down + 1, cols] = 1
            if 0 <= inner_up < height:       # bottom bar
                hori[inner_up:min(height - 1, box.down) + 1, cols] = 1
    return vert, hori


def rectangle_streak_image(config: RectangleStreakConfig, state, image: Image, rng: Optional[RandomGenerator]):
    _check_color(image, config.color)
    aspect_ratio = config.aspect_ratio
    if aspect_ratio is None:
        aspect_ratio = image.width / image.height
    boxes = generate_centered_boxes(image.height, image.width, aspect_ratio, config.short_side_min,
                                    config.short_side_step)
    vert, hori = _rectangle_bar_masks(image.height, image.width, boxes, config.thickness)
    if config.dash_thickness > 0 and config.dash_gap > 0:
        step = config.dash_thickness + config.dash_gap
        for offset in range(config.dash_gap):
            vert[offset::step] = 0
            hori[:, offset::step] = 0
    out = np.array(image.mat, order='C')
    cn = out.shape[2]
    box = (0, 0, image.height, image.width)
    _native.fill(out, [
        _native.make_layer(box, cn, tuple(config.color), mask=vert, alpha=config.alpha),
        _native.make_layer(box, cn, tuple(config.color), mask=hori, alpha=config.alpha),
    ])
    return attrs.evolve(image, mat=out)


rectangle_streak = Distortion(
    config_cls=RectangleStreakConfig,
    state_cls=DistortionNopState[RectangleStreakConfig],
    func_image=rectangle_streak_image,
)


@attrs.define
class EllipseStreakConfig(DistortionConfig):
    thickness: int = 1
    aspect_ratio: Optional[float] = None
    short_side_min: int = 10
    short_side_step: int = 10
    color: Tuple[int, int, int] = (0, 0, 0)
    alpha: float = 1.0


def ellipse_streak_image(config: EllipseStreakConfig, state, image: Image, rng: Optional[RandomGenerator]):
    """reference photometric/streak.py:296-330: one ``cv.ellipse`` outline per concentric box (axes = half the box sides,
    centre = the image centre), blended like the other streaks.  Raster and blend run on the device in one call
    (csrc/ellipse.hip)."""
    _check_color(image, config.color)
    if not isinstance(config.alpha, float):
        raise AttributeError('alpha must be a float')
    if config.alpha < 0.0 or config.alpha > 1.0:
        raise RuntimeError(f'alpha={config.alpha} is invalid.')
    aspect_ratio = config.aspect_ratio
    if aspect_ratio is None:
        aspect_ratio = image.width / image.height
    boxes = generate_centered_boxes(image.height, image.width, aspect_ratio, config.short_side_min,
                                    config.short_side_step)
    axes = [(box.width // 2, box.height // 2) for box in boxes]
    mat = _native.ellipse_streak(image.arr, (image.width // 2, image.height // 2), axes, config.thickness,
                                 config.color, config.alpha)
    return attrs.evolve(image, mat=mat)


ellipse_streak = Distortion(
    config_cls=EllipseStreakConfig,
    state_cls=DistortionNopState[EllipseStreakConfig],
    func_image=ellipse_streak_image,
)
