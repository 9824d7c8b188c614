"""Binary mask element: uint8 HxW, active where > 0 (reference: vkit/element/mask.py)."""
from typing import Iterable, Optional, Tuple, Union

import attrs
import numpy as np

from ._writable import LazyMat, WritableContext
from .opt import generate_resized_shape
from .type import ElementSetOperationMode, Shapable


_LUT_INVERT = (np.arange(256) == 0).astype(np.uint8).reshape(1, 256)           # (~(mat > 0)).astype(uint8)
_LUT_X255 = ((np.arange(256) > 0).astype(np.uint8) * 255).reshape(1, 256)      # (mat > 0).astype(uint8) * 255


@attrs.define(frozen=True, eq=False)
class Mask(LazyMat, Shapable):
    _mat: np.ndarray = attrs.field(alias='mat')
    box: Optional['Box'] = None

    _np_mask: Optional[np.ndarray] = attrs.field(default=None, init=False, repr=False)

    def __attrs_post_init__(self):
        if self._mat.dtype != np.uint8:
            raise RuntimeError('mat.dtype != np.uint8')
        if self._mat.ndim != 2:
            raise RuntimeError('ndim should == 2.')
        if isinstance(self._mat, np.ndarray):
            self._mat.flags.writeable = False
        if self.box and self.shape != self.box.shape:
            raise RuntimeError('self.shape != box.shape.')

    # ---- constructors
    @classmethod
    def from_shape(cls, shape: Tuple[int, int], value: int = 0):
        height, width = shape
        assert value in (0, 1)
        init = np.zeros if value == 0 else np.ones
        return cls(mat=init((height, width), dtype=np.uint8))

    @classmethod
    def from_shapable(cls, shapable: Shapable, value: int = 0):
        return cls.from_shape(shapable.shape, value=value)

    @classmethod
    def _from_np_active_count(cls, shape, mode, np_active_count, attached_box):
        if mode == ElementSetOperationMode.UNION:
            active = np_active_count > 0
        elif mode == ElementSetOperationMode.DISTINCT:
            active = np_active_count == 1
        elif mode == ElementSetOperationMode.INTERSECT:
            active = np_active_count > 1
        else:
            raise NotImplementedError()
        mask = cls(mat=active.astype(np.uint8))
        return mask.to_box_attached(attached_box) if attached_box else mask

    @classmethod
    def from_boxes(cls, shape_or_box, boxes: Iterable['Box'],
                   mode: ElementSetOperationMode = ElementSetOperationMode.UNION):
        attached_box = shape_or_box if isinstance(shape_or_box, Box) else None
        shape = attached_box.shape if attached_box else shape_or_box
        count = np.zeros(shape, dtype=np.int32)
        for box in boxes:
            if attached_box:
                box = box.to_relative_box(origin_y=attached_box.up, origin_x=attached_box.left)
            box.extract_np_array(count)[...] += 1
        return cls._from_np_active_count(shape, mode, count, attached_box)

    # ---- properties
    @property
    def height(self):
        return self._mat.shape[0]

    @property
    def width(self):
        return self._mat.shape[1]

    @property
    def equivalent_box(self):
        return self.box or Box.from_shapable(self)

    @property
    def np_mask(self):
        if self._np_mask is None:
            object.__setattr__(self, '_np_mask', self.mat > 0)
        return self._np_mask

    def set_np_mask_out_of_date(self):
        object.__setattr__(self, '_np_mask', None)

    @property
    def writable_context(self):
        return WritableContext(self, on_exit=self.set_np_mask_out_of_date)

    # ---- operators
    def copy(self):
        return attrs.evolve(self, mat=self.mat.copy())

    def assign_mat(self, mat: np.ndarray):
        with self.writable_context:
            object.__setattr__(self, '_mat', mat)

    def to_inverted_mask(self):
        if self.on_device:
            from vkit_amd import _native
            return attrs.evolve(self, mat=_native.apply_lut(self.arr, _LUT_INVERT))
        return attrs.evolve(self, mat=(~self.np_mask).astype(np.uint8))

    def to_shifted_mask(self, offset_y: int = 0, offset_x: int = 0):
        assert self.box
        return attrs.evolve(self, box=self.box.to_shifted_box(offset_y=offset_y, offset_x=offset_x))

    def to_resized_mask(self, resized_height: Optional[int] = None, resized_width: Optional[int] = None,
                        cv_resize_interpolation: int = 2, binarization_threshold: int = 0):
        """cv.resize of the 0/255 plane (any cv2 code 0..6), then ``> binarization_threshold`` (reference mask.py:454-479)."""
        from vkit_amd import _native
        assert not self.box
        if cv_resize_interpolation not in range(7):
            raise ValueError(f'unknown cv2 interpolation code {cv_resize_interpolation}')
        resized_height, resized_width = generate_resized_shape(
            height=self.height, width=self.width, resized_height=resized_height, resized_width=resized_width)
        if self.on_device or _native.resident_mode():
            # the same three steps as table look-ups around the device resize: (> 0) * 255, resize, > threshold
            plane = _native.apply_lut(self.arr, _LUT_X255)
            plane = _native.resize(plane, (resized_height, resized_width), cv_resize_interpolation)
            above = (np.arange(256) > binarization_threshold).astype(np.uint8).reshape(1, 256)
            return Mask(mat=_native.apply_lut(plane, above))
        mat = _native.resize(self.np_mask.astype(np.uint8) * 255, (resized_height, resized_width), cv_resize_interpolation)
        return Mask(mat=(mat > binarization_threshold).astype(np.uint8))

    def to_conducted_resized_mask(self, shapable_or_shape, resized_height: Optional[int] = None,
                                  resized_width: Optional[int] = None, cv_resize_interpolation: int = 2,
                                  binarization_threshold: int = 0):
        assert self.box
        resized_box = self.box.to_conducted_resized_box(shapable_or_shape=shapable_or_shape,
                                                        resized_height=resized_height, resized_width=resized_width)
        resized_mask = self.to_box_detached().to_resized_mask(
            resized_height=resized_box.height, resized_width=resized_box.width,
            cv_resize_interpolation=cv_resize_interpolation, binarization_threshold=binarization_threshold)
        return resized_mask.to_box_attached(resized_box)

    def to_cropped_mask(self, up=None, down=None, left=None, right=None):
        assert not self.box
        up = up or 0
        down = down or self.height - 1
        left = left or 0
        right = right or self.width - 1
        return attrs.evolve(self, mat=self.mat[up:down + 1, left:right + 1])

    def to_box_attached(self, box: 'Box'):
        assert self.height == box.height and self.width == box.width
        return attrs.evolve(self, box=box)

    def to_box_detached(self):
        assert self.box
        return attrs.evolve(self, box=None)

    def to_score_map(self):
        return ScoreMap(mat=self.np_mask.astype(np.float32), box=self.box)

    # ---- fills (self selects the pixels)
    def fill_np_array(self, mat: np.ndarray, value, alpha=1.0, keep_max_value: bool = False,
                      keep_min_value: bool = False):
        self.equivalent_box.fill_np_array(mat, value, np_mask=self.np_mask, alpha=alpha,
                                          keep_max_value=keep_max_value, keep_min_value=keep_min_value)

    def fill_mask(self, mask: 'Mask', value: Union['Mask', np.ndarray, int] = 1, keep_max_value: bool = False,
                  keep_min_value: bool = False):
        self.equivalent_box.fill_mask(mask, value, mask_mask=self, keep_max_value=keep_max_value,
                                      keep_min_value=keep_min_value)

    def fill_score_map(self, score_map: 'ScoreMap', value, keep_max_value: bool = False,
                       keep_min_value: bool = False):
        self.equivalent_box.fill_score_map(score_map, value, score_map_mask=self, keep_max_value=keep_max_value,
                                           keep_min_value=keep_min_value)

    def fill_image(self, image: 'Image', value, alpha: Union['ScoreMap', np.ndarray, float] = 1.0):
        self.equivalent_box.fill_image(image, value, image_mask=self, alpha=alpha)

    def extract_mask(self, mask: 'Mask'):
        out = self.equivalent_box.extract_mask(mask).copy()
        self.to_inverted_mask().fill_mask(out, value=0)
        return out

    def extract_image(self, image: 'Image'):
        out = self.equivalent_box.extract_image(image).copy()
        self.to_inverted_mask().fill_image(out, value=0)
        return out


def generate_fill_by_masks_mask(shape, masks, mode: ElementSetOperationMode):
    if mode == ElementSetOperationMode.UNION:
        return None
    raise NotImplementedError('non-UNION mask set operations are outside the accelerated path')


from .box import Box  # noqa: E402
from .score_map import ScoreMap  # noqa: E402
from .image import Image  # noqa: E402
