"""``fog`` (reference: photometric/effect.py:89-216).  The fog density field is a diamond-square fractal drawn from the
caller-visible numpy Generator stream -- on the device for PCG64 (``_native.np_fog_mask``: numpy's float32 / float64 roundings level
by level), with numpy itself otherwise (``generate_diamond_square_mask``, the statement-by-statement restatement); the per-pixel work -- blend
every pixel towards the fog colour with the field as float32 alpha, ``uint8(clip((1 - a) * px + a * fog))`` -- is the
alpha composite ``vkx_fill_u8`` with one page-sized layer.  ``pixelation`` (:56-86) shrinks with ``cv.resize``
INTER_LINEAR and grows back with INTER_NEAREST (``vkx_resize_u8``).  ``jpeg_quality`` (:25-53, an encoder round trip
through ``cv.imencode`` / ``cv.imdecode``) is outside the path: the operator exists with the reference's config so that
policies sample it draw for draw, the image passes through (``photometric/opt.py: pass_through_out_of_path``)."""
from typing import Any, Mapping, Optional, Tuple

import attrs
import numpy as np
from numpy.random import Generator as RandomGenerator

from vkit_amd import _native
from vkit_amd.element import Image, ImageMode
from ..interface import Distortion, DistortionConfig, DistortionNopState
from .opt import pass_through_out_of_path


@attrs.define
class JpegQualityConfig(DistortionConfig):
    quality: int


def jpeg_quality_image(config: JpegQualityConfig, state, image: Image, rng: Optional[RandomGenerator]):
    assert 0 <= config.quality <= 100
    return pass_through_out_of_path('jpeg_quality', image)


jpeg_quality = Distortion(
    config_cls=JpegQualityConfig,
    state_cls=DistortionNopState[JpegQualityConfig],
    func_image=jpeg_quality_image,
)


@attrs.define
class PixelationConfig(DistortionConfig):
    ratio: float


def pixelation_image(config: PixelationConfig, state, image: Image, rng: Optional[RandomGenerator]):
    assert 0 < config.ratio < 1
    small_shape = (round(image.height * config.ratio), round(image.width * config.ratio))
    small = _native.resize(image.arr, small_shape, _native.INTER_LINEAR)
    return attrs.evolve(image, mat=_native.resize(small, image.shape, _native.INTER_NEAREST))


pixelation = Distortion(
    config_cls=PixelationConfig,
    state_cls=DistortionNopState[PixelationConfig],
    func_image=pixelation_image,
)


def generate_diamond_square_mask(shape: Tuple[int, int], roughness: float, rng: RandomGenerator):
    """Midpoint-displacement field on a (2^k + 1)^2 lattice, cropped at a random offset.  Draw order and arithmetic follow
    the reference statement by statement: four corner draws; per level the diamond centres, then the two families of
    edge midpoints, each ``(1 - r^level) * neighbour_sum / 4 + r^level * uniform`` with wrap-around neighbours."""
    assert 0.0 <= roughness <= 1.0
    height, width = shape
    size = int(2**np.ceil(np.log2(max(height, width))) + 1)
    field = np.zeros((size, size), dtype=np.float32)
    field[0, 0] = rng.uniform(0.0, 1.0)
    field[0, -1] = rng.uniform(0.0, 1.0)
    field[-1, -1] = rng.uniform(0.0, 1.0)
    field[-1, 0] = rng.uniform(0.0, 1.0)

    step, level = size - 1, 0
    while step >= 2:
        noise_weight = roughness**level
        half = step // 2
        corners = field[0:size:step, 0:size:step]
        pair_down = corners + np.roll(corners, shift=-1, axis=0)    # corner + the one below (wraps)
        pair_right = corners + np.roll(corners, shift=-1, axis=1)   # corner + the one to the right (wraps)

        # centres of the squares
        around = (pair_down + pair_right)[:-1, :-1]
        centres = (1 - noise_weight) * around / 4 + noise_weight * rng.uniform(0, 1, around.shape)
        field[half:size:step, half:size:step] = centres

        # midpoints of the horizontal edges: the two corners of the edge + the centres above and below
        centres_vert = centres + np.roll(centres, shift=1, axis=0)
        centres_vert = np.vstack([centres_vert, centres_vert[0]])
        around = pair_right[:, :-1] + centres_vert
        field[0:size:step, half:size:step] = ((1 - noise_weight) * around / 4
                                              + noise_weight * rng.uniform(0, 1, around.shape))

        # midpoints of the vertical edges
        centres_hori = centres + np.roll(centres, shift=1, axis=1)
        centres_hori = np.hstack([centres_hori, centres_hori[0].reshape(-1, 1)])
        around = pair_down[:-1] + centres_hori
        field[half:size:step, 0:size:step] = ((1 - noise_weight) * around / 4
                                              + noise_weight * rng.uniform(0, 1, around.shape))
        level += 1
        step = half

    up = rng.integers(0, size - height + 1)
    left = rng.integers(0, size - width + 1)
    return field[up:up + height, left:left + width]


@attrs.define
class FogConfig(DistortionConfig):
    roughness: float
    fog_rgb: Tuple[int, int, int] = (226, 238, 234)
    ratio_max: float = 1.0
    ratio_min: float = 0.0

    _rng_state: Optional[Mapping[str, Any]] = None

    @property
    def supports_rng_state(self) -> bool:
        return True

    @property
    def rng_state(self) -> Optional[Mapping[str, Any]]:
        return self._rng_state

    @rng_state.setter
    def rng_state(self, val: Mapping[str, Any]):
        self._rng_state = val


def fog_image(config: FogConfig, state, image: Image, rng: Optional[RandomGenerator]):
    mode = image.mode
    if mode not in (ImageMode.GRAYSCALE, ImageMode.RGB):
        image = image.to_rgb_image()
    assert rng is not None
    assert config.ratio_min < config.ratio_max
    # the lattice levels and the stretch on the device, from the caller's PCG64 stream (vkx_fog_field_f32_dev); None: numpy below
    mask = _native.np_fog_mask(image.shape, config.roughness, config.ratio_min, config.ratio_max, rng)
    on_device = mask is not None      # a DevArray: it stays in HBM as the alpha plane of the blend
    if not on_device:
        mask = generate_diamond_square_mask(image.shape, config.roughness, rng)
        # stretch the field to [ratio_min, ratio_max] (float32 in place, like the reference)
        mask = np.array(mask, dtype=np.float32)
        mask -= mask.min()
        mask /= mask.max()
        mask *= (config.ratio_max - config.ratio_min)
        mask += config.ratio_min

    if image.mode == ImageMode.GRAYSCALE:
        # the grey fog value is fractional (reference effect.py:194-197): float32(0.2126 R + 0.7152 G + 0.0722 B)
        val = 0.2126 * config.fog_rgb[0] + 0.7152 * config.fog_rgb[1] + 0.0722 * config.fog_rgb[2]
        return attrs.evolve(image, mat=_native.fog_f32(image.arr, mask, [np.float32(val)]))
    if on_device:
        mat = _native.device_copy(image.arr, mask.ctx)
    else:
        mat = np.array(image.mat)
    layer = _native.make_layer((0, 0, image.height, image.width), 3, tuple(int(v) for v in config.fog_rgb), alpha=mask)
    _native.fill(mat, [layer])
    if on_device and not _native.resident_mode() and not isinstance(image.arr, _native.DevArray):
        mat = np.array(mat.host())        # host in, host out
    image = attrs.evolve(image, mat=mat)
    if mode != ImageMode.RGB:
        image = image.to_target_mode_image(mode)
    return image


fog = Distortion(
    config_cls=FogConfig,
    state_cls=DistortionNopState[FogConfig],
    func_image=fog_image,
)
