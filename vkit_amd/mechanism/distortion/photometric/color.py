"""Colour distortions on the accelerated path: ``mean_shift``, ``color_shift`` (reference:
photometric/color.py:32-116) and the integer per-value members ``complement``, ``posterization``,
``channel_permutation`` (:299-357, :400-432), ``brightness_shift`` (:125-160), ``color_balance`` (:360-397) and the two
equalisations (:205-285): a per-channel histogram on the GPU (exact integer reduction), a 256-entry table per channel
built on the host with the reference's arithmetic, and a table pass on the GPU.  ``std_shift`` (:165-210) takes its
per-channel float32 mean -- an order-dependent float32 reduction -- from ``vkx_sum_f32_u8``, which adds in numpy's order with
numpy's roundings (csrc/reduce.hip), and then is a table pass like the others."""
from typing import Any, Mapping, Optional, Sequence

import os

import attrs
import numpy as np
from numpy.random import Generator as RandomGenerator

from vkit_amd import _native
from vkit_amd.element import Image, ImageMode
from ..interface import Distortion, DistortionConfig, DistortionNopState
from .opt import OutOfBoundBehavior


def _mean_shift(image: Image, channels: Optional[Sequence[int]], delta: int, threshold: Optional[int],
                oob_behavior: OutOfBoundBehavior):
    """int16(px) + delta on the selected channels (optionally gated by a threshold), then CLIP or CYCLE to uint8."""
    if delta == 0:
        return image
    if oob_behavior not in (OutOfBoundBehavior.CLIP, OutOfBoundBehavior.CYCLE):
        raise NotImplementedError()
    if threshold is not None:
        assert delta != 0
    mat = _native.mean_shift(image.arr, delta, threshold=threshold, channels=channels,
                             cycle=(oob_behavior == OutOfBoundBehavior.CYCLE))
    return attrs.evolve(image, mat=mat)


@attrs.define
class MeanShiftConfig(DistortionConfig):
    delta: int
    threshold: Optional[int] = None
    channels: Optional[Sequence[int]] = None
    oob_behavior: OutOfBoundBehavior = OutOfBoundBehavior.CLIP


def mean_shift_image(config: MeanShiftConfig, state, image: Image, rng: Optional[RandomGenerator]):
    return _mean_shift(image, config.channels, config.delta, config.threshold, config.oob_behavior)


mean_shift = Distortion(
    config_cls=MeanShiftConfig,
    state_cls=DistortionNopState[MeanShiftConfig],
    func_image=mean_shift_image,
)


@attrs.define
class ColorShiftConfig(DistortionConfig):
    delta: int


def color_shift_image(config: ColorShiftConfig, state, image: Image, rng: Optional[RandomGenerator]):
    """Rotate the hue by ``delta`` / 256 of a turn.  RGB images take the fused HIP kernel (RGB -> HSV_FULL,
    H += delta mod 256, HSV_FULL -> RGB in one pass); HSV / HSL images only need the hue add."""
    mode = image.mode
    if mode in (ImageMode.HSV, ImageMode.HSL):
        return _mean_shift(image, [0], config.delta, None, OutOfBoundBehavior.CYCLE)
    if mode != ImageMode.RGB:
        # GRAYSCALE / RGBA: the reference's own route (color.py:93-116): to HSV, hue add, back to the mode
        shifted = _mean_shift(image.to_hsv_image(), [0], config.delta, None, OutOfBoundBehavior.CYCLE)
        return shifted.to_target_mode_image(mode)
    if config.delta == 0:
        # the reference still round-trips through HSV (the hue add is skipped, the conversions are not)
        return image.to_hsv_image().to_target_mode_image(mode)
    return Image(mat=_native.color_shift_rgb(image.arr, config.delta), mode=ImageMode.RGB)


color_shift = Distortion(
    config_cls=ColorShiftConfig,
    state_cls=DistortionNopState[ColorShiftConfig],
    func_image=color_shift_image,
)


@attrs.define
class ComplementConfig(DistortionConfig):
    threshold: Optional[int] = None
    enable_threshold_lte: bool = False
    channels: Optional[Sequence[int]] = None


def complement_image(config: ComplementConfig, state, image: Image, rng: Optional[RandomGenerator]):
    """255 - v on the selected channels, optionally only where ``threshold <= v`` (or ``v <= threshold``)."""
    if config.threshold is not None:
        assert 0 <= config.threshold <= 255
    mat = _native.pointwise(image.arr, _native.POINT_COMPLEMENT,
                            -1 if config.threshold is None else int(config.threshold),
                            int(config.enable_threshold_lte), channels=config.channels)
    return attrs.evolve(image, mat=mat)


complement = Distortion(
    config_cls=ComplementConfig,
    state_cls=DistortionNopState[ComplementConfig],
    func_image=complement_image,
)


@attrs.define
class PosterizationConfig(DistortionConfig):
    num_bits: int
    channels: Optional[Sequence[int]] = None


def posterization_image(config: PosterizationConfig, state, image: Image, rng: Optional[RandomGenerator]):
    """Clears the lower ``num_bits`` bits of the selected channels."""
    assert 0 <= config.num_bits < 8
    if config.num_bits == 0:
        return image
    mat = _native.pointwise(image.arr, _native.POINT_POSTERIZE, int(config.num_bits), channels=config.channels)
    return attrs.evolve(image, mat=mat)


posterization = Distortion(
    config_cls=PosterizationConfig,
    state_cls=DistortionNopState[PosterizationConfig],
    func_image=posterization_image,
)


@attrs.define
class ChannelPermutationConfig(DistortionConfig):
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


def channel_permutation_image(config: ChannelPermutationConfig, state, image: Image,
                              rng: Optional[RandomGenerator]):
    """``mat[:, :, rng.permutation(num_channels)]``: the permutation is drawn on the host, the gather runs on the GPU."""
    assert rng
    indices = rng.permutation(image.num_channels)
    return attrs.evolve(image, mat=_native.permute_channels(image.arr, indices))


channel_permutation = Distortion(
    config_cls=ChannelPermutationConfig,
    state_cls=DistortionNopState[ChannelPermutationConfig],
    func_image=channel_permutation_image,
)


@attrs.define
class BrightnessShiftConfig(DistortionConfig):
    delta: int
    intermediate_image_mode: ImageMode = ImageMode.HSL


def brightness_shift_image(config: BrightnessShiftConfig, state, image: Image, rng: Optional[RandomGenerator]):
    """Adds ``delta`` to the lightness (HSL) / value (HSV) channel with clipping.  An RGB image with the default HSL
    intermediate takes one fused kernel (RGB -> HLS_FULL, L += delta, HLS_FULL -> RGB)."""
    mode = image.mode
    if mode == ImageMode.RGB and config.intermediate_image_mode == ImageMode.HSL:
        return Image(mat=_native.brightness_shift_rgb(image.arr, config.delta), mode=ImageMode.RGB)
    if mode not in (ImageMode.HSV, ImageMode.HSL):
        assert config.intermediate_image_mode in (ImageMode.HSV, ImageMode.HSL)
        image = image.to_target_mode_image(config.intermediate_image_mode)
    image = _mean_shift(image, [2], config.delta, None, OutOfBoundBehavior.CLIP)
    if mode not in (ImageMode.HSV, ImageMode.HSL):
        image = image.to_target_mode_image(mode)
    return image


brightness_shift = Distortion(
    config_cls=BrightnessShiftConfig,
    state_cls=DistortionNopState[BrightnessShiftConfig],
    func_image=brightness_shift_image,
)


@attrs.define
class ColorBalanceConfig(DistortionConfig):
    ratio: float


def color_balance_image(config: ColorBalanceConfig, state, image: Image, rng: Optional[RandomGenerator]):
    """Blend towards the grey version of the image: ``(1 - ratio) * gray + ratio * px`` in float32, truncated."""
    if image.mode == ImageMode.GRAYSCALE:
        return image
    assert 0.0 <= config.ratio <= 1.0
    if image.mode == ImageMode.RGB:
        return attrs.evolve(image, mat=_native.color_balance_rgb(image.arr, config.ratio))
    # any other mode (reference color.py:380-396): the grey version of the image brought back to the image's mode, blended
    # with the image -- on saturation and value / lightness only for HSV / HSL, on every channel (alpha included) for RGBA
    grayscale_like = image.to_grayscale_image().to_target_mode_image(image.mode)
    channels = [1, 2] if image.mode in (ImageMode.HSV, ImageMode.HSL) else None
    mat = _native.blend_u8(grayscale_like.arr, image.arr, 1 - config.ratio, config.ratio, channels=channels)
    return attrs.evolve(image, mat=mat)


color_balance = Distortion(
    config_cls=ColorBalanceConfig,
    state_cls=DistortionNopState[ColorBalanceConfig],
    func_image=color_balance_image,
)


@attrs.define
class StdShiftConfig(DistortionConfig):
    scale: float
    channels: Optional[Sequence[int]] = None


def std_shift_image(config: StdShiftConfig, state, image: Image, rng: Optional[RandomGenerator]):
    """``round(v * scale - mean * (scale - 1))`` clipped to uint8, per selected channel (reference color.py:165-203).

    The mean is the reference's expression, ``np.mean`` of the float32 copy -- over the flattened pixels with ``axis=0`` for colour
    images: a float32 accumulation whose value depends on the order of the additions.  ``vkx_sum_f32_u8`` forms it on the device in
    numpy's order, rounding for rounding (csrc/reduce.hip; 5 ms of a host core per 1024^2 page before); images beyond its limits
    take numpy.  Everything per pixel depends on the grey level and the channel only: the float32 expression is evaluated for the
    256 levels and applied as a table on the GPU."""
    assert config.scale > 0
    ndim = image.arr.ndim
    mean = None
    if os.environ.get('VKX_HOST_MEAN') != '1' and ndim in (2, 3):
        mean = _native.mean_f32_u8(image.arr, list(config.channels) if (config.channels and ndim == 3) else None)
    if mean is None:
        mat = image.mat[:, :, list(config.channels)] if config.channels else image.mat
        mat = mat.astype(np.float32)
        if mat.ndim == 2:
            mean = np.mean(mat)
        elif mat.ndim == 3:
            mean = np.mean(mat.reshape(-1, mat.shape[-1]), axis=0)
        else:
            raise NotImplementedError()
    selected = _selected_channels(image, config.channels)
    levels = np.arange(256, dtype=np.float32)
    if ndim == 3:
        values = levels.reshape(-1, 1) * config.scale - mean * (config.scale - 1)       # (256, k), float32
    else:
        values = (levels * config.scale - mean * (config.scale - 1)).reshape(-1, 1)
    table = np.clip(np.round(values), 0, 255).astype(np.uint8)
    lut = np.tile(np.arange(256, dtype=np.uint8), (max(image.num_channels, 1), 1))
    for k, c in enumerate(selected):
        lut[c] = table[:, k]
    return attrs.evolve(image, mat=_native.apply_lut(image.arr, lut, channels=selected))


std_shift = Distortion(
    config_cls=StdShiftConfig,
    state_cls=DistortionNopState[StdShiftConfig],
    func_image=std_shift_image,
)


def _selected_channels(image: Image, channels: Optional[Sequence[int]]):
    return list(channels) if channels else list(range(max(image.num_channels, 1)))


@attrs.define
class BoundaryEqualizationConfig(DistortionConfig):
    channels: Optional[Sequence[int]] = None


def boundary_equalization_image(config: BoundaryEqualizationConfig, state, image: Image,
                                rng: Optional[RandomGenerator]):
    """Stretch every selected channel to [0, 255]: ``round((v - min) * (255 / (max - min)))`` in float32.  The minimum /
    maximum come from the GPU histogram; the per-value expression is evaluated once per grey level into a table."""
    hist = _native.histogram(image.arr)
    selected = _selected_channels(image, config.channels)
    lut = np.tile(np.arange(256, dtype=np.uint8), (hist.shape[0], 1))
    levels = np.arange(256, dtype=np.float32)
    any_delta = False
    for c in selected:
        present = np.nonzero(hist[c])[0]
        if present.size == 0:
            continue
        vmin, vmax = np.float32(present[0]), np.float32(present[-1])
        delta = vmax - vmin
        if not delta > 0:
            continue            # a flat channel is left alone (reference :232-240)
        any_delta = True
        # 3-D images divide a python float by a float32 array, 2-D ones by a float32 scalar: float32 either way
        values = (levels - vmin) * (np.float32(255.0) / delta)
        lut[c] = np.clip(np.round(values), 0, 255).astype(np.uint8)
    if not any_delta:
        return image
    return attrs.evolve(image, mat=_native.apply_lut(image.arr, lut, channels=selected))


boundary_equalization = Distortion(
    config_cls=BoundaryEqualizationConfig,
    state_cls=DistortionNopState[BoundaryEqualizationConfig],
    func_image=boundary_equalization_image,
)


@attrs.define
class HistogramEqualizationConfig(DistortionConfig):
    channels: Optional[Sequence[int]] = None


def _equalize_hist_table(hist_c: np.ndarray) -> np.ndarray:
    """cv.equalizeHist's table from one channel histogram (imgproc/histogram.cpp): cumulative count above the first
    occupied bin times 255 / (total - count of that bin), float32, rounded half to even, saturated."""
    lut = np.arange(256, dtype=np.uint8)
    present = np.nonzero(hist_c)[0]
    total = int(hist_c.sum())
    if present.size == 0 or int(hist_c[present[0]]) == total:
        return lut                                           # single-valued plane: unchanged
    first = int(present[0])
    scale = np.float32(255.0) / np.float32(total - int(hist_c[first]))
    cum = np.cumsum(hist_c.astype(np.int64))
    acc = (cum - cum[first]).astype(np.float32)            # sum of bins first+1 .. i
    table = np.clip(np.rint(acc * scale), 0, 255).astype(np.uint8)
    table[:first + 1] = 0
    return table


def histogram_equalization_image(config: HistogramEqualizationConfig, state, image: Image,
                                 rng: Optional[RandomGenerator]):
    hist = _native.histogram(image.arr)
    selected = _selected_channels(image, config.channels)
    lut = np.tile(np.arange(256, dtype=np.uint8), (hist.shape[0], 1))
    for c in selected:
        lut[c] = _equalize_hist_table(hist[c])
    return attrs.evolve(image, mat=_native.apply_lut(image.arr, lut, channels=selected))


histogram_equalization = Distortion(
    config_cls=HistogramEqualizationConfig,
    state_cls=DistortionNopState[HistogramEqualizationConfig],
    func_image=histogram_equalization_image,
)
