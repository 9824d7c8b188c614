"""Axis-aligned integer box (reference: vkit/element/box.py).  ``down`` / ``right`` are inclusive.

``Box.fill_*`` is the entry of the alpha composite used by page assembly
(vkit/pipeline/text_detection/page_assembler.py:155-236): the box selects the destination window, the optional
mask / alpha plane select and weight pixels, and the blend runs on the GPU (``vkx_fill_u8``).
"""
import math
from typing import Iterable, Optional, Tuple, Union

import attrs
import numpy as np

from vkit_amd import _native

from .opt import (
    DeviceWindow,
    _deferred_stack,
    clip_val,
    extract_shape_from_shapable_or_shape,
    fill_np_array,
    generate_shape_and_resized_shape,
    resize_val,
)
from .type import ElementSetOperationMode, Shapable


@attrs.define(frozen=True)
class Box(Shapable):
    up: int
    down: int
    left: int
    right: int

    # ---- constructors
    @classmethod
    def from_shape(cls, shape: Tuple[int, int]):
        height, width = shape
        return cls(up=0, down=height - 1, left=0, right=width - 1)

    @classmethod
    def from_shapable(cls, shapable: Shapable):
        return cls.from_shape(shapable.shape)

    @classmethod
    def from_boxes(cls, boxes: Iterable['Box']):
        boxes = list(boxes)
        return cls(
            up=min(b.up for b in boxes),
            down=max(b.down for b in boxes),
            left=min(b.left for b in boxes),
            right=max(b.right for b in boxes),
        )

    # ---- properties
    @property
    def height(self):
        return self.down + 1 - self.up

    @property
    def width(self):
        return self.right + 1 - self.left

    @property
    def valid(self):
        return (0 <= self.up <= self.down) and (0 <= self.left <= self.right)

    # ---- conversion
    def to_polygon(self, step: Optional[int] = None):
        if self.up == self.down or self.left == self.right:
            raise RuntimeError(f'Cannot convert box={self} to polygon.')
        if step is None:
            # up-left, up-right, down-right, down-left (char-level labels rely on this order)
            points = PointTuple.from_xy_pairs((
                (self.left, self.up), (self.right, self.up), (self.right, self.down), (self.left, self.down)))
        else:
            assert step > 0
            xs = list(range(self.left, self.right + 1, step))
            if xs[-1] < self.right:
                xs.append(self.right)
            ys = list(range(self.up, self.down + 1, step))
            if ys[-1] == self.down:
                ys.pop()
            ys.pop(0)
            points = PointList()
            points.extend(Point.create(y=self.up, x=x) for x in xs)
            points.extend(Point.create(y=y, x=self.right) for y in ys)
            points.extend(Point.create(y=self.down, x=x) for x in reversed(xs))
            points.extend(Point.create(y=y, x=self.left) for y in reversed(ys))
        return Polygon.create(points=points)

    # ---- operators
    def get_center_point(self):
        return Point.create(y=(self.up + self.down) / 2, x=(self.left + self.right) / 2)

    def to_clipped_box(self, shapable_or_shape: Union[Shapable, Tuple[int, int]]):
        height, width = extract_shape_from_shapable_or_shape(shapable_or_shape)
        return Box(up=clip_val(self.up, height), down=clip_val(self.down, height),
                   left=clip_val(self.left, width), right=clip_val(self.right, width))

    def to_conducted_resized_box(self, shapable_or_shape, resized_height: Optional[int] = None,
                                 resized_width: Optional[int] = None):
        height, width, resized_height, resized_width = generate_shape_and_resized_shape(
            shapable_or_shape, resized_height, resized_width)
        return Box(
            up=round(resize_val(self.up, height, resized_height)),
            down=round(resize_val(self.down, height, resized_height)),
            left=round(resize_val(self.left, width, resized_width)),
            right=round(resize_val(self.right, width, resized_width)),
        )

    def to_resized_box(self, resized_height: Optional[int] = None, resized_width: Optional[int] = None):
        return self.to_conducted_resized_box(self, resized_height=resized_height, resized_width=resized_width)

    def to_shifted_box(self, offset_y: int = 0, offset_x: int = 0):
        return Box(up=self.up + offset_y, down=self.down + offset_y, left=self.left + offset_x,
                   right=self.right + offset_x)

    def to_relative_box(self, origin_y: int, origin_x: int):
        return self.to_shifted_box(offset_y=-origin_y, offset_x=-origin_x)

    def to_dilated_box(self, ratio: float, clip_long_side: bool = False):
        expand_vert = math.ceil(self.height * ratio / 2)
        expand_hori = math.ceil(self.width * ratio / 2)
        if clip_long_side:
            expand_vert = expand_hori = min(expand_vert, expand_hori)
        return Box(up=self.up - expand_vert, down=self.down + expand_vert, left=self.left - expand_hori,
                   right=self.right + expand_hori)

    def get_boxes_for_box_attached_opt(self, element_box: Optional['Box']):
        """(box relative to the element's own array, box to attach to an extracted element)."""
        if element_box is None:
            return self, None
        assert element_box.up <= self.up <= self.down <= element_box.down
        assert element_box.left <= self.left <= self.right <= element_box.right
        return self.to_relative_box(origin_y=element_box.up, origin_x=element_box.left), self

    def extract_np_array(self, mat: np.ndarray) -> np.ndarray:
        assert 0 <= self.up <= self.down <= mat.shape[0]
        assert 0 <= self.left <= self.right <= mat.shape[1]
        return mat[self.up:self.down + 1, self.left:self.right + 1]

    def _extract_element(self, element):
        relative_box, new_box = self.get_boxes_for_box_attached_opt(element.box)
        if relative_box.shape == element.shape:
            return element
        return attrs.evolve(element, mat=relative_box.extract_np_array(element.mat), box=new_box)

    def extract_mask(self, mask: 'Mask'):
        return self._extract_element(mask)

    def extract_score_map(self, score_map: 'ScoreMap'):
        return self._extract_element(score_map)

    def extract_image(self, image: 'Image'):
        return self._extract_element(image)

    @classmethod
    def get_np_mask_from_element_mask(cls, element_mask):
        if element_mask is None:
            return None
        if isinstance(element_mask, Mask):
            return element_mask.np_mask  # Mask.box is ignored
        return element_mask

    def fill_np_array(
        self,
        mat: np.ndarray,
        value,
        np_mask: Optional[np.ndarray] = None,
        alpha: Union['ScoreMap', np.ndarray, float] = 1.0,
        keep_max_value: bool = False,
        keep_min_value: bool = False,
    ):
        full_shape = (mat.shape[0], mat.shape[1])
        if full_shape == self.shape:
            window = mat
        elif isinstance(mat, np.ndarray):
            window = self.extract_np_array(mat)
        else:       # a device-resident array: the composite takes the box as layer geometry, nothing is sliced
            window = DeviceWindow(self.shape + tuple(mat.shape[2:]), mat.dtype)
        if isinstance(value, np.ndarray):
            if (value.shape[0], value.shape[1]) != (window.shape[0], window.shape[1]):
                assert (value.shape[0], value.shape[1]) == full_shape
                value = self.extract_np_array(value)
            if value.dtype != mat.dtype:
                value = value.astype(mat.dtype)
        if isinstance(alpha, ScoreMap):
            assert alpha.is_prob  # ScoreMap.box is ignored
            alpha = alpha.mat
        if np_mask is None and isinstance(alpha, np.ndarray) and (keep_max_value or keep_min_value or mat.dtype != np.uint8):
            # sparse alpha: untouched where alpha == 0.  A plain uint8 blend needs no selection plane for that:
            # (1 - 0) * d + 0 * v is d exactly in float32, so the composite leaves those pixels as they are by arithmetic
            np_mask = (alpha > 0.0)
        origin = None if window is mat else (mat, self.up, self.left)
        fill_np_array(window, value, np_mask=np_mask, alpha=alpha, keep_max_value=keep_max_value,
                      keep_min_value=keep_min_value, origin=origin)

    def _fill_element(self, element, value, value_cls, element_mask, **kwargs):
        relative_box, _ = self.get_boxes_for_box_attached_opt(element.box)
        if isinstance(value, value_cls):
            if value.shape != self.shape:
                value = self._extract_element(value)
            value = value.mat
        np_mask = self.get_np_mask_from_element_mask(element_mask)
        stack = _deferred_stack()
        if not isinstance(element._mat, np.ndarray):
            # device-resident element: no write flag to manage; recorded by an open deferred composite or applied now
            relative_box.fill_np_array(element._mat, value, np_mask=np_mask, **kwargs)
            return
        if stack and stack[-1].base is element._mat:
            # an open deferred composite on this element records the layer; it writes (and handles the write flag) on exit
            relative_box.fill_np_array(element._mat, value, np_mask=np_mask, **kwargs)
            return
        with element.writable_context:
            relative_box.fill_np_array(element.mat, value, np_mask=np_mask, **kwargs)

    def fill_mask(self, mask: 'Mask', value: Union['Mask', np.ndarray, int] = 1, mask_mask=None,
                  keep_max_value: bool = False, keep_min_value: bool = False):
        self._fill_element(mask, value, Mask, mask_mask, keep_max_value=keep_max_value,
                           keep_min_value=keep_min_value)

    def fill_score_map(self, score_map: 'ScoreMap', value: Union['ScoreMap', np.ndarray, float],
                       score_map_mask=None, keep_max_value: bool = False, keep_min_value: bool = False):
        self._fill_element(score_map, value, ScoreMap, score_map_mask, keep_max_value=keep_max_value,
                           keep_min_value=keep_min_value)

    def fill_image(self, image: 'Image', value, image_mask=None, alpha: Union['ScoreMap', np.ndarray, float] = 1.0):
        # The layers of a page recorded by an open deferred composite on a uint8 image without a box of its own: the record
        # the generic path below builds (same geometry, planes and checks), built directly.
        target = image._mat
        if image.box is None and target.dtype == np.uint8 and target.ndim == 3:
            stack = _deferred_stack()
            if (stack and stack[-1].base is target and 0 <= self.up <= self.down < target.shape[0]
                    and 0 <= self.left <= self.right < target.shape[1]):
                shape = self.shape
                plane = value._mat if isinstance(value, Image) else value
                ok = isinstance(plane, tuple) or (isinstance(plane, np.ndarray) and plane.dtype == np.uint8
                                                 and plane.shape == shape + (target.shape[2],))
                mask_plane = None
                if image_mask is not None:
                    mask_plane = image_mask._mat if isinstance(image_mask, Mask) else image_mask
                    ok = ok and isinstance(mask_plane, np.ndarray) and mask_plane.shape == shape and mask_plane.dtype in (np.uint8, np.bool_)
                    if ok and mask_plane.dtype == np.bool_:
                        mask_plane = mask_plane.view(np.uint8)
                weight = alpha
                if isinstance(alpha, ScoreMap):
                    ok = ok and alpha.is_prob
                    weight = alpha._mat
                if isinstance(weight, np.ndarray):
                    ok = ok and weight.shape == shape and weight.dtype == np.float32
                else:
                    ok = ok and type(weight) is float and 0.0 <= weight <= 1.0
                if ok:
                    stack[-1].layers.append(_native.make_layer((self.up, self.left) + shape, target.shape[2], plane, mask=mask_plane,
                                                               alpha=weight))
                    return
        self._fill_element(image, value, Image, image_mask, alpha=alpha)


def generate_fill_by_boxes_mask(shape: Tuple[int, int], boxes: Iterable[Box], mode: ElementSetOperationMode):
    if mode == ElementSetOperationMode.UNION:
        return None
    return Mask.from_boxes(shape, boxes, mode)


# Cyclic by design, like the reference.
from .point import Point, PointList, PointTuple  # noqa: E402
from .polygon import Polygon  # noqa: E402
from .mask import Mask  # noqa: E402
from .score_map import ScoreMap  # noqa: E402
from .image import Image  # noqa: E402
