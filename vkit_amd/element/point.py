"""Point value types (reference: vkit/element/point.py).

A ``Point`` carries the smooth (float) position used by geometric distortions and the integer pixel
position ``round(smooth)`` (Python's round: half to even).  Equality and hashing use the integer position
only.  ``PointTuple.to_smooth_np_array`` deliberately returns the *integer* positions as float32, like the
reference does (vkit/element/point.py:251-252); MLS handles and camera strategies depend on that.
"""
from itertools import chain
from typing import Iterable, List, Optional, Sequence, Tuple, Union

import attrs
import numpy as np

from .opt import clip_val, extract_shape_from_shapable_or_shape, generate_shape_and_resized_shape, resize_val
from .type import Shapable

_Num = Union[float, str]


@attrs.define(frozen=True)
class Point:
    smooth_y: float = attrs.field(eq=False)
    smooth_x: float = attrs.field(eq=False)
    y: int = attrs.field(init=False, hash=False)
    x: int = attrs.field(init=False, hash=False)

    def __attrs_post_init__(self):
        object.__setattr__(self, 'y', round(self.smooth_y))
        object.__setattr__(self, 'x', round(self.smooth_x))

    @classmethod
    def create(cls, y: _Num, x: _Num):
        return cls(smooth_y=float(y), smooth_x=float(x))

    @classmethod
    def from_xy_pair(cls, xy_pair: Tuple[_Num, _Num]):
        x, y = xy_pair
        return cls.create(y=y, x=x)

    def to_xy_pair(self):
        return self.x, self.y

    def to_smooth_xy_pair(self):
        return self.smooth_x, self.smooth_y

    def to_clipped_point(self, shapable_or_shape: Union[Shapable, Tuple[int, int]]):
        height, width = extract_shape_from_shapable_or_shape(shapable_or_shape)
        if 0 <= self.y < height and 0 <= self.x < width:
            return self
        return Point.create(y=clip_val(self.smooth_y, height), x=clip_val(self.smooth_x, width))

    def to_shifted_point(self, offset_y: int = 0, offset_x: int = 0):
        return Point.create(y=self.smooth_y + offset_y, x=self.smooth_x + offset_x)

    def to_conducted_resized_point(self, shapable_or_shape, resized_height: Optional[int] = None,
                                   resized_width: Optional[int] = None):
        height, width, resized_height, resized_width = generate_shape_and_resized_shape(
            shapable_or_shape, resized_height, resized_width)
        return Point.create(
            y=resize_val(self.smooth_y, height, resized_height),
            x=resize_val(self.smooth_x, width, resized_width),
        )


class _PointSeqMixin:
    """Shared behaviour of PointList / PointTuple; ``_wrap`` rebuilds the concrete container."""

    def to_xy_pairs(self):
        return self._seq(p.to_xy_pair() for p in self)

    def to_smooth_xy_pairs(self):
        return self._seq(p.to_smooth_xy_pair() for p in self)

    def to_flatten_xy_pairs(self):
        return self._seq(chain.from_iterable(p.to_xy_pair() for p in self))

    def to_smooth_flatten_xy_pairs(self):
        return self._seq(chain.from_iterable(p.to_smooth_xy_pair() for p in self))

    def to_np_array(self):
        return np.asarray([p.to_xy_pair() for p in self], dtype=np.int32)

    def to_clipped_points(self, shapable_or_shape):
        return self._wrap(p.to_clipped_point(shapable_or_shape) for p in self)

    def to_shifted_points(self, offset_y: int = 0, offset_x: int = 0):
        return self._wrap(p.to_shifted_point(offset_y=offset_y, offset_x=offset_x) for p in self)

    def to_relative_points(self, origin_y: int, origin_x: int):
        return self.to_shifted_points(offset_y=-origin_y, offset_x=-origin_x)

    def to_conducted_resized_points(self, shapable_or_shape, resized_height: Optional[int] = None,
                                    resized_width: Optional[int] = None):
        return self._wrap(
            p.to_conducted_resized_point(shapable_or_shape, resized_height=resized_height, resized_width=resized_width)
            for p in self)


class PointList(_PointSeqMixin, List[Point]):
    _seq = staticmethod(list)

    @classmethod
    def _wrap(cls, it):
        return cls(it)

    @classmethod
    def from_point(cls, point: Point):
        return cls((point,))

    @classmethod
    def from_xy_pairs(cls, xy_pairs: Iterable[Tuple[_Num, _Num]]):
        return cls(Point.from_xy_pair(pair) for pair in xy_pairs)

    @classmethod
    def from_flatten_xy_pairs(cls, flatten_xy_pairs: Sequence[_Num]):
        flat = tuple(flatten_xy_pairs)
        assert flat and len(flat) % 2 == 0
        return cls(Point.create(y=flat[i + 1], x=flat[i]) for i in range(0, len(flat), 2))

    @classmethod
    def from_np_array(cls, np_points: np.ndarray):
        points = cls(Point.create(y=row[1], x=row[0]) for row in np_points)
        # a closing duplicate (as produced by shapely) is dropped
        if len(points) > 2 and points[0] == points[-1]:
            points.pop()
        return points

    def to_smooth_np_array(self):
        return np.asarray([p.to_smooth_xy_pair() for p in self], dtype=np.float32)

    def to_point_tuple(self):
        return PointTuple(self)

    def copy(self):
        return PointList(self)


class PointTuple(_PointSeqMixin, Tuple[Point, ...]):
    _seq = staticmethod(tuple)

    @classmethod
    def _wrap(cls, it):
        return cls(it)

    @classmethod
    def from_point(cls, point: Point):
        return cls((point,))

    @classmethod
    def from_xy_pairs(cls, xy_pairs: Iterable[Tuple[_Num, _Num]]):
        return cls(Point.from_xy_pair(pair) for pair in xy_pairs)

    @classmethod
    def from_flatten_xy_pairs(cls, flatten_xy_pairs: Sequence[_Num]):
        return PointList.from_flatten_xy_pairs(flatten_xy_pairs).to_point_tuple()

    @classmethod
    def from_np_array(cls, np_points: np.ndarray):
        return PointList.from_np_array(np_points).to_point_tuple()

    def to_smooth_np_array(self):
        # NOTE: integer positions on purpose (reference quirk, see module docstring).
        return np.asarray([p.to_xy_pair() for p in self], dtype=np.float32)
