"""``gaussian_blur`` (reference: photometric/blur.py:26-69): ksize = max(3, round(3 sigma) + 1) forced odd,
``cv.GaussianBlur(mat, (k, k), sigma)`` -- OpenCV's bit-exact 8.8 fixed-point separable kernel restated in HIP --
and ``glass_blur`` (:186-258): that blur followed by a random local pixel shuffle.  ``defocus_blur`` / ``motion_blur``
(:85-192) build a small float32 kernel on the host -- a normalised disc, or a line rotated with ``cv.warpAffine``
(``vkx_warp_affine_f32``), smoothed with a float32 ``cv.GaussianBlur`` restated in numpy below -- and apply it with
``cv.filter2D`` on the GPU (``vkx_filter2d_u8``; the scalar C++ arithmetic of cv2: float32 accumulation in tap order,
no fused multiply-add).  ``zoom_in_blur`` (:264-323) averages the image with centred crops of its bicubic
enlargements (``vkx_zoom_in_blur_u8``)."""
import math
import os
from typing import Any, Mapping, Optional

import attrs
import numpy as np
from numpy.random import Generator as RandomGenerator

from vkit_amd import _native
from vkit_amd.element import Image
from ..interface import Distortion, DistortionConfig, DistortionNopState
from .opt import to_original_image, to_rgb_image


def _estimate_gaussian_kernel_size(sigma: float):
    kernel_size = max(3, round(3 * sigma) + 1)
    return kernel_size + 1 if kernel_size % 2 == 0 else kernel_size


@attrs.define
class GaussianBlurConfig(DistortionConfig):
    sigma: float


def gaussian_blur_image(config: GaussianBlurConfig, state, image: Image, rng: Optional[RandomGenerator]):
    mode = image.mode
    image = to_rgb_image(image, mode)
    mat = _native.gaussian_blur(image.arr, _estimate_gaussian_kernel_size(config.sigma), config.sigma)
    return to_original_image(attrs.evolve(image, mat=mat), mode)


gaussian_blur = Distortion(
    config_cls=GaussianBlurConfig,
    state_cls=DistortionNopState[GaussianBlurConfig],
    func_image=gaussian_blur_image,
)


def _gaussian_kernel_f32(ksize: int, sigma: float) -> np.ndarray:
    """cv.getGaussianKernel(ksize, sigma, CV_32F) for sigma > 0: the bit-exact double kernel (weights exp(-x^2 / 2 sigma^2)
    evaluated on doubled integer offsets, normalised by the reciprocal of their sum), cast to float32."""
    assert ksize % 2 == 1 and sigma > 0
    scale2x = -0.125 / (sigma * sigma)
    half = (ksize - 1) // 2
    values = [math.exp(float(x * x) * scale2x) for x in range(1 - ksize, 0, 2)]
    total = 0.0
    for v in values:
        total += v
    total *= 2
    total += 1
    mul1 = 1.0 / total
    kernel = np.empty(ksize, np.float32)
    for i, v in enumerate(values):
        kernel[i] = kernel[ksize - 1 - i] = np.float32(v * mul1)
    kernel[half] = np.float32(1.0 * mul1)
    return kernel


def _reflect101(idx: np.ndarray, size: int) -> np.ndarray:
    if size == 1:
        return np.zeros_like(idx)
    idx = np.abs(idx)
    period = 2 * (size - 1)
    idx = idx % period
    return np.where(idx >= size, period - idx, idx)


def _gaussian_blur_f32(mat: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    """cv.GaussianBlur on a small float32 matrix (the kernels below): separable, rows then columns, the symmetric form
    ``x0 k0 + (x-1 + x+1) k1 + ...`` in float32, BORDER_REFLECT_101."""
    kernel = _gaussian_kernel_f32(ksize, sigma)
    r = ksize // 2
    out = mat.astype(np.float32)
    for axis in (1, 0):
        n = out.shape[axis]
        base = np.arange(n)
        acc = out * kernel[r]
        for j in range(1, r + 1):
            pair = np.take(out, _reflect101(base - j, n), axis=axis) + np.take(out, _reflect101(base + j, n), axis=axis)
            acc = acc + pair * kernel[r + j]
        out = acc.astype(np.float32)
    return out


def _anti_aliasing_ksize_and_padding(anti_aliasing_sigma: float):
    kernel_size = _estimate_gaussian_kernel_size(anti_aliasing_sigma)
    return kernel_size, kernel_size // 2 * 2


def _filter2d_image(image: Image, kernel: np.ndarray):
    mode = image.mode
    image = to_rgb_image(image, mode)
    return to_original_image(attrs.evolve(image, mat=_native.filter2d(image.arr, kernel)), mode)


@attrs.define
class DefocusBlurConfig(DistortionConfig):
    radius: int
    anti_aliasing_sigma: float = 0.5


def defocus_kernel(radius: int, anti_aliasing_sigma: float) -> np.ndarray:
    """Normalised disc of the given radius with a zero margin, smoothed by the anti-aliasing Gaussian (reference :90-112)."""
    assert 0 < radius
    aa_ksize, padding = _anti_aliasing_ksize_and_padding(anti_aliasing_sigma)
    kernel_size = 2 * radius + 1 + padding
    begin = -(kernel_size // 2)
    coords = np.arange(begin, begin + kernel_size)
    x, y = np.meshgrid(coords, coords)
    kernel = ((x ** 2 + y ** 2) <= radius ** 2).astype(np.float32)
    kernel /= kernel.sum()
    return _gaussian_blur_f32(kernel, aa_ksize, anti_aliasing_sigma)


def defocus_blur_image(config: DefocusBlurConfig, state, image: Image, rng: Optional[RandomGenerator]):
    return _filter2d_image(image, defocus_kernel(config.radius, config.anti_aliasing_sigma))


defocus_blur = Distortion(
    config_cls=DefocusBlurConfig,
    state_cls=DistortionNopState[DefocusBlurConfig],
    func_image=defocus_blur_image,
)


@attrs.define
class MotionBlurConfig(DistortionConfig):
    radius: int
    angle: int
    anti_aliasing_sigma: float = 0.5


def _rotation_matrix_2d(center, angle: float, scale: float) -> np.ndarray:
    """cv.getRotationMatrix2D: degrees, counter-clockwise, float64."""
    angle = angle * np.pi / 180
    alpha, beta = np.cos(angle) * scale, np.sin(angle) * scale
    return np.asarray([[alpha, beta, (1 - alpha) * center[0] - beta * center[1]],
                       [-beta, alpha, beta * center[0] + (1 - alpha) * center[1]]], np.float64)


def motion_kernel(radius: int, angle: int, anti_aliasing_sigma: float) -> np.ndarray:
    """A horizontal line of 2 radius + 1 taps in a zero margin, rotated clockwise by ``angle`` degrees with a bilinear
    warpAffine, normalised, smoothed (reference :135-176)."""
    aa_ksize, padding = _anti_aliasing_ksize_and_padding(anti_aliasing_sigma)
    half = padding // 2
    center, left = radius + half, half
    length = 2 * radius + 1
    kernel_size = length + padding
    kernel = np.zeros((kernel_size, kernel_size), np.float32)
    kernel[center, left:left + length] = 1.0
    trans_mat = _rotation_matrix_2d((center, center), 360 - (int(angle) % 360), 1.0)
    # a kernel of a few dozen taps that numpy goes on with: a host array whatever mode the caller runs in
    with _native.resident(False):
        kernel = np.array(_native.host_array(_native.warp_affine(kernel, trans_mat, kernel.shape)))
    kernel /= kernel.sum()
    return _gaussian_blur_f32(kernel, aa_ksize, anti_aliasing_sigma)


def motion_blur_image(config: MotionBlurConfig, state, image: Image, rng: Optional[RandomGenerator]):
    return _filter2d_image(image, motion_kernel(config.radius, config.angle, config.anti_aliasing_sigma))


motion_blur = Distortion(
    config_cls=MotionBlurConfig,
    state_cls=DistortionNopState[MotionBlurConfig],
    func_image=motion_blur_image,
)


@attrs.define
class GlassBlurConfig(DistortionConfig):
    sigma: float
    delta: int = 1
    loop: int = 5

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


def glass_shuffle_planes(shape, delta: int, loop: int, rng: RandomGenerator):
    """The (row, column) source planes of the glass shuffle.  Per round: a lattice of cell centres with a random phase
    and pitch 2 delta + 1; every centre swaps its CURRENT source position with that of a random neighbour within delta
    (clipped to the image).  Index bookkeeping on the caller's rng stream; numpy's assignment order (later duplicates
    win) is part of the result, so the swaps are written as the reference writes them."""
    height, width = shape
    pos_x, pos_y = np.meshgrid(np.arange(width), np.arange(height))
    pitch = 2 * delta + 1
    for _ in range(loop):
        rows = np.arange(rng.integers(0, pitch), height - delta, pitch).reshape(-1, 1)
        cols = np.arange(rng.integers(0, pitch), width - delta, pitch).reshape(1, -1)
        grid = (rows.shape[0], cols.shape[1])
        jump_y = rng.integers(-delta, delta + 1, grid)
        jump_x = rng.integers(-delta, delta + 1, grid)
        to_y = np.clip(pos_y[rows, cols] + jump_y, 0, height - 1)
        to_x = np.clip(pos_x[rows, cols] + jump_x, 0, width - 1)
        pos_y[rows, cols], pos_y[to_y, to_x] = pos_y[to_y, to_x], pos_y[rows, cols]
        pos_x[rows, cols], pos_x[to_y, to_x] = pos_x[to_y, to_x], pos_x[rows, cols]
    return pos_y, pos_x


def glass_blur_image(config: GlassBlurConfig, state, image: Image, rng: Optional[RandomGenerator]):
    mode = image.mode
    image = to_rgb_image(image, mode)
    assert rng is not None
    mat = _native.gaussian_blur(image.arr, _estimate_gaussian_kernel_size(config.sigma), config.sigma)
    # the shuffle planes stay on the device: the jumps are rng's draws, the swap rounds run there (VKX_HOST_SHUFFLE=1: numpy's planes)
    if os.environ.get('VKX_HOST_SHUFFLE', '') == '1' or max(image.shape) > 32767:
        pos_y, pos_x = glass_shuffle_planes(image.shape, config.delta, config.loop, rng)
    else:
        pos_y, pos_x = _native.glass_shuffle_planes_dev(image.shape, config.delta, config.loop, rng)
    mat = _native.gather(mat, pos_y, pos_x)
    return to_original_image(attrs.evolve(image, mat=mat), mode)


glass_blur = Distortion(
    config_cls=GlassBlurConfig,
    state_cls=DistortionNopState[GlassBlurConfig],
    func_image=glass_blur_image,
)


@attrs.define
class ZoomInBlurConfig(DistortionConfig):
    ratio: float = 0.1
    step: float = 0.01
    alpha: float = 0.5


def zoom_in_blur_image(config: ZoomInBlurConfig, state, image: Image, rng: Optional[RandomGenerator]):
    mode = image.mode
    image = to_rgb_image(image, mode)
    # the enlargement factors 1 + step, 1 + 2 step, ... up to 1 + ratio (numpy's float arange, like the reference)
    sizes = [(round(image.height * factor), round(image.width * factor))
             for factor in np.arange(1 + config.step, 1 + config.ratio + config.step, config.step)]
    mat = _native.zoom_in_blur(image.arr, sizes, config.alpha)
    return to_original_image(attrs.evolve(image, mat=mat), mode)


zoom_in_blur = Distortion(
    config_cls=ZoomInBlurConfig,
    state_cls=DistortionNopState[ZoomInBlurConfig],
    func_image=zoom_in_blur_image,
)

