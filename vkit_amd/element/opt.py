"""Scalar helpers of the element layer and the uint8 alpha-composite entry point.

``fill_np_array`` keeps the reference's signature and semantics (vkit/element/opt.py:118-209) but the
arithmetic runs in the ``k_fill`` HIP kernel through ``vkx_fill_u8``; there is no numpy fallback.
"""
import threading
from typing import Optional, Tuple, Union

import numpy as np

from vkit_amd import _native
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


class deferred_fill:
    """Collects every ``fill_np_array`` aimed at ``base`` inside the ``with`` block and applies them, in call order,
    with ONE ``vkx_fill_u8`` call on exit: the destination crosses PCIe once per page instead of once per layer
    (the page assembler's loop, reference pipeline/text_detection/page_assembler.py:150-236).

    Layer sources (value / mask / alpha arrays) must stay unchanged until the block exits, and nothing may read
    ``base`` inside the block -- the writes have not happened yet.
    """

    def __init__(self, base):
        """``base``: a numpy array, or the ``DevArray`` of a device-resident element (the layers' host planes are staged
        for the one call, the page never leaves the device)."""
        self.on_device = isinstance(base, _native.DevArray)
        if base.dtype not in (np.uint8, np.float32) or not (self.on_device or base.flags.c_contiguous):
            raise ValueError('deferred_fill needs a C-contiguous uint8 or float32 destination')
        self.base = base
        self.layers = []

    def accepts(self, base: np.ndarray):
        return base is self.base

    def __enter__(self):
        _deferred_stack().append(self)
        return self

    def __exit__(self, exc_type, exc, tb):
        stack = _deferred_stack()
        if self in stack:
            stack.remove(self)
        # an exception inside the block abandons the recorded layers: the destination is left as it was, the
        # exception propagates
        if exc_type is None and self.layers and self.on_device:
            _native.fill(self.base, self.layers)
        elif exc_type is None and self.layers:
            writeable = self.base.flags.writeable
            self.base.flags.writeable = True
            try:
                _native.fill(self.base, self.layers)
            finally:
                self.base.flags.writeable = writeable
        self.layers = []
        return False


class DeviceWindow:
    """What ``fill_np_array`` needs to know of a box of a device-resident array: shape, dtype, ndim."""
    __slots__ = ('shape', 'dtype', 'ndim')

    def __init__(self, shape, dtype):
        self.shape, self.dtype, self.ndim = tuple(shape), np.dtype(dtype), len(shape)


# one stack per thread: contexts (streams, scratch) are per thread too (_native.default_ctx), and two threads
# assembling pages at once must not see each other's open composites
_DEFERRED_TLS = threading.local()


def _deferred_stack():
    stack = getattr(_DEFERRED_TLS, 'stack', None)
    if stack is None:
        stack = _DEFERRED_TLS.stack = []
    return stack


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
    if mat.dtype == np.float32:
        if mat.ndim != 2:
            raise NotImplementedError('float32 fills are implemented for 2-D arrays (ScoreMap)')
    elif mat.dtype != np.uint8:
        raise NotImplementedError(f'fill_np_array on dtype {mat.dtype} is outside the accelerated path.')
    mode = _native.FILL_PLAIN
    if keep_max_value or keep_min_value:
        assert not (keep_max_value and keep_min_value)
        mode = _native.FILL_KEEP_MAX if keep_max_value else _native.FILL_KEEP_MIN
    if not isinstance(alpha, (float, np.ndarray)):
        # the reference fails here too (an int alpha misses both isinstance checks, vkit/element/opt.py:128,146,195)
        raise AttributeError(f'alpha must be a float or a numpy array, got {type(alpha).__name__}')
    if isinstance(alpha, float) and (alpha < 0.0 or alpha > 1.0):
        raise RuntimeError(f'alpha={alpha} is invalid.')
    if isinstance(alpha, float) and 0.0 < alpha < 1.0 and np_mask is not None and mat.ndim == 2:
        # the reference indexes shape[1] of the boolean-indexed (1-D) selection here (vkit/element/opt.py:172-176)
        raise IndexError('tuple index out of range')

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
    layer = _native.make_layer((up, left, bh, bw), cn, value, mask=mask_u8, alpha=alpha, mode=mode, dtype=mat.dtype)
    stack = _deferred_stack()
    if stack and stack[-1].accepts(base):
        stack[-1].layers.append(layer)
        return
    if isinstance(base, _native.DevArray) or base.flags.c_contiguous:
        _native.fill(base, [layer])
    else:
        packed = np.ascontiguousarray(base)
        _native.fill(packed, [layer])
        base[...] = packed
