"""Scalar helpers of the element layer and the uint8 alpha-composite entry point.

``fill_np_array`` keeps the reference's signature and semantics (vkit/element/opt.py:118-209) but the
arithmetic runs in the ``k_fill`` HIP kernel through ``vkx_fill_u8``; there is no numpy fallback.
"""
from typing import Optional, Tuple, Union

import numpy as np

from .type import Shapable


def clip_val(val, size: int):
    return max(0, min(val, size - 1))


def resize_val(val, size: int, resized_size: int):
    return clip_val(val * resized_size / size, resized_size)


def extract_shape_from_shapable_or_shape(shapable_or_shape: Union[Shapable, Tuple[int, int]]):
    if isinstance(shapable_or_shape, Shapable):
        return shapable_or_shape.shape
    height, width = shapable_or_shape
    return height, width


def generate_resized_shape(height: int, width: int, resized_height: Optional[int] = None,
                           resized_width: Optional[int] = None):
    if not resized_height and not resized_width:
        raise RuntimeError('Missing resized_height or resized_width.')
    if resized_height is None:
        resized_height = round(resized_width * height / width)
    if resized_width is None:
        resized_width = round(resized_height * width / height)
    return resized_height, resized_width


def generate_shape_and_resized_shape(shapable_or_shape, resized_height: Optional[int] = None,
                                     resized_width: Optional[int] = None):
    height, width = extract_shape_from_shapable_or_shape(shapable_or_shape)
    resized_height, resized_width = generate_resized_shape(height, width, resized_height, resized_width)
    return height, width, resized_height, resized_width


def fill_np_array(
    mat: np.ndarray,
    value,
    np_mask: Optional[np.ndarray] = None,
    alpha: Union[np.ndarray, float] = 1.0,
    keep_max_value: bool = False,
    keep_min_value: bool = False,
    origin: Optional[Tuple[np.ndarray, int, int]] = None,
):
    """Masked / alpha-weighted write of ``value`` into ``mat`` (in place) on the GPU.

    ``mat`` may be a (non-contiguous) box view of a larger array; pass ``origin=(base, up, left)`` so the
    composite is issued on the contiguous base array with the box as layer geometry.
    """
    from vkit_amd import _native

    if keep_max_value or keep_min_value:
        raise NotImplementedError('keep_max_value / keep_min_value fills (label rasterisation) are not on the '
                                  'accelerated path yet.')
    if mat.dtype != np.uint8:
        raise NotImplementedError(f'fill_np_array on dtype {mat.dtype} is not on the accelerated path yet.')
    if not isinstance(alpha, (float, np.ndarray)):
        # the reference fails here too (an int alpha misses both isinstance checks, vkit/element/opt.py:128,146,195)
        raise AttributeError(f'alpha must be a float or a numpy array, got {type(alpha).__name__}')
    if isinstance(alpha, float) and (alpha < 0.0 or alpha > 1.0):
        raise RuntimeError(f'alpha={alpha} is invalid.')

    if origin is None:
        base, up, left = mat, 0, 0
    else:
        base, up, left = origin
    bh, bw = mat.shape[:2]
    cn = 1 if mat.ndim == 2 else mat.shape[2]
    if isinstance(value, np.ndarray):
        if value.shape != mat.shape:
            raise RuntimeError('value is np.ndarray but shape is not matched.')
    elif mat.ndim == 2 and isinstance(value, tuple):
        raise ValueError('a tuple value needs a 3-D destination')
    mask_u8 = None
    if np_mask is not None:
        mask_u8 = np_mask.view(np.uint8) if np_mask.dtype == np.bool_ else (np_mask > 0).view(np.uint8)
    layer = _native.make_layer((up, left, bh, bw), cn, value, mask=mask_u8, alpha=alpha)
    if base.flags.c_contiguous:
        _native.fill(base, [layer])
    else:
        packed = np.ascontiguousarray(base)
        _native.fill(packed, [layer])
        base[...] = packed
