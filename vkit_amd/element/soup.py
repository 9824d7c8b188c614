"""Array-native point and polygon containers.

The reference moves points and polygons through its distortion chain as Python objects: a page of 64 text lines and
384 char boxes is ~1 800 polygon vertices + ~900 height points, and every geometric operator, the clipping after it
and the final trim rebuild each of them (``Point.create`` -> two ``round`` calls -> attrs ``__init__``;
distortion/interface.py:638-661, element/point.py:31-120).  On this path that object traffic was 12 of the 20 ms of a
``PageDistortionStep.run`` whose device work is 0.4 ms.

``PointArray`` and ``PolygonSoup`` hold the same information as one float64 array of smooth (x, y) positions (plus
vertex offsets for the soup) and are *sequences* of ``Point`` / ``Polygon``: indexing and iteration materialise the
objects on demand (cached), so callers written against the reference's containers keep working, while the operators
of the path (grid projection, affine transforms, clipping, shifting, resizing, label painting) run on the arrays and
never build an object.  The arithmetic is the reference's, element for element:

* integer position = ``round(smooth)`` with Python's round-half-to-even  ==  ``np.rint`` on float64;
* ``to_clipped_point`` (point.py:57-65): a point whose INTEGER position is inside keeps its smooth position, any
  other point has both smooth coordinates clamped to [0, size - 1];
* ``to_conducted_resized_point`` (:73-88): ``clip(val * resized / size, resized)`` in that operation order;
* ``PointTuple.to_smooth_np_array`` returns the integer positions as float32, ``PointList.to_smooth_np_array`` the
  smooth ones (point.py:251-252 quirk): ``tuple_like`` selects which;
* ``Polygon.from_np_array`` / ``PointList.from_np_array`` drop a closing duplicate (first == last by integer position,
  more than two points): ``PolygonSoup.from_np_arrays_dropping_closing_duplicates`` does the same per polygon.
"""
from typing import Iterable, Optional, Sequence, Tuple

import numpy as np

from .opt import extract_shape_from_shapable_or_shape, generate_shape_and_resized_shape


def _rint_xy(smooth_xy: np.ndarray) -> np.ndarray:
    return np.rint(smooth_xy).astype(np.int64)


def _clip_to_shape(smooth_xy: np.ndarray, int_xy: np.ndarray, shape: Tuple[int, int]) -> np.ndarray:
    height, width = shape
    x, y = int_xy[:, 0], int_xy[:, 1]
    outside = (y < 0) | (y >= height) | (x < 0) | (x >= width)
    if not outside.any():
        return smooth_xy
    out = smooth_xy.copy()
    # + 0.0: Python's max(0, -0.0) is +0
    out[outside, 0] = np.clip(smooth_xy[outside, 0], 0, width - 1) + 0.0
    out[outside, 1] = np.clip(smooth_xy[outside, 1], 0, height - 1) + 0.0
    return out


def _resize_xy(smooth_xy: np.ndarray, shape_and_resized) -> np.ndarray:
    height, width, resized_height, resized_width = shape_and_resized
    out = np.empty_like(smooth_xy)
    out[:, 0] = np.clip(smooth_xy[:, 0] * resized_width / width, 0, resized_width - 1) + 0.0
    out[:, 1] = np.clip(smooth_xy[:, 1] * resized_height / height, 0, resized_height - 1) + 0.0
    return out


class PointArray(Sequence):
    """A sequence of ``Point`` backed by one float64 (n, 2) array of smooth (x, y)."""

    __slots__ = ('smooth_xy', 'tuple_like', '_int_xy', '_points')

    def __init__(self, smooth_xy: np.ndarray, tuple_like: bool = False):
        smooth_xy = np.asarray(smooth_xy, dtype=np.float64).reshape(-1, 2)
        self.smooth_xy = smooth_xy
        self.tuple_like = tuple_like
        self._int_xy: Optional[np.ndarray] = None
        self._points = None

    # ---- construction
    @classmethod
    def from_points(cls, points: Iterable, tuple_like: bool = False):
        if isinstance(points, PointArray):
            return points if points.tuple_like == tuple_like else points._retagged(tuple_like)
        pts = list(points)
        arr = np.empty((len(pts), 2), np.float64)
        for k, p in enumerate(pts):
            arr[k, 0] = p.smooth_x
            arr[k, 1] = p.smooth_y
        out = cls(arr, tuple_like)
        out._points = pts
        return out

    def _retagged(self, tuple_like: bool):
        out = PointArray(self.smooth_xy, tuple_like)
        out._int_xy, out._points = self._int_xy, self._points
        return out

    # ---- arrays
    @property
    def int_xy(self) -> np.ndarray:
        if self._int_xy is None:
            self._int_xy = _rint_xy(self.smooth_xy)
        return self._int_xy

    def to_np_array(self):
        return self.int_xy.astype(np.int32)

    def to_smooth_np_array(self):
        return (self.int_xy if self.tuple_like else self.smooth_xy).astype(np.float32)

    # ---- sequence protocol
    def __len__(self):
        return self.smooth_xy.shape[0]

    def _materialised(self):
        if self._points is None:
            from .point import Point
            self._points = [Point.create(y=float(y), x=float(x)) for x, y in self.smooth_xy]
        return self._points

    def __getitem__(self, index):
        if isinstance(index, slice):
            out = PointArray(self.smooth_xy[index], self.tuple_like)
            if self._int_xy is not None:
                out._int_xy = self._int_xy[index]
            return out
        return self._materialised()[index]

    def __iter__(self):
        return iter(self._materialised())

    def __eq__(self, other):
        if isinstance(other, PointArray):
            return self.int_xy.shape == other.int_xy.shape and bool((self.int_xy == other.int_xy).all())
        try:
            return len(other) == len(self) and all(a == b for a, b in zip(self, other))
        except TypeError:
            return NotImplemented

    __hash__ = None

    def __repr__(self):
        return f'PointArray({len(self)} points, {"tuple" if self.tuple_like else "list"}-like)'

    # ---- the containers' operators, vectorised
    def to_point_tuple(self):
        return self._retagged(True)

    def to_point_list(self):
        return self._retagged(False)

    def copy(self):
        return self._retagged(self.tuple_like)

    def to_xy_pairs(self):
        return [tuple(int(v) for v in row) for row in self.int_xy]

    def to_smooth_xy_pairs(self):
        return [tuple(float(v) for v in row) for row in self.smooth_xy]

    def to_clipped_points(self, shapable_or_shape):
        shape = extract_shape_from_shapable_or_shape(shapable_or_shape)
        clipped = _clip_to_shape(self.smooth_xy, self.int_xy, shape)
        return self if clipped is self.smooth_xy else PointArray(clipped, self.tuple_like)

    def to_shifted_points(self, offset_y: int = 0, offset_x: int = 0):
        return PointArray(self.smooth_xy + np.array([offset_x, offset_y], np.float64), self.tuple_like)

    def to_relative_points(self, origin_y: int, origin_x: int):
        return self.to_shifted_points(offset_y=-origin_y, offset_x=-origin_x)

    def to_conducted_resized_points(self, shapable_or_shape, resized_height: Optional[int] = None,
                                    resized_width: Optional[int] = None):
        sizes = generate_shape_and_resized_shape(shapable_or_shape, resized_height, resized_width)
        return PointArray(_resize_xy(self.smooth_xy, sizes), self.tuple_like)


class PolygonSoup(Sequence):
    """A sequence of ``Polygon`` backed by one float64 (N, 2) array of smooth (x, y) vertices and (P + 1) offsets."""

    __slots__ = ('smooth_xy', 'offsets', '_int_xy', '_polygons')

    def __init__(self, smooth_xy: np.ndarray, offsets: np.ndarray):
        self.smooth_xy = np.asarray(smooth_xy, dtype=np.float64).reshape(-1, 2)
        self.offsets = np.asarray(offsets, dtype=np.int64)
        assert self.offsets[0] == 0 and self.offsets[-1] == self.smooth_xy.shape[0]
        self._int_xy: Optional[np.ndarray] = None
        self._polygons = None

    # ---- construction
    @classmethod
    def from_polygons(cls, polygons: Iterable):
        if isinstance(polygons, PolygonSoup):
            return polygons
        polygons = list(polygons)
        arrays = [polygon.smooth_xy for polygon in polygons]
        offsets = np.zeros(len(arrays) + 1, np.int64)
        if arrays:
            offsets[1:] = np.cumsum([a.shape[0] for a in arrays])
        out = cls(np.concatenate(arrays, axis=0) if arrays else np.zeros((0, 2), np.float64), offsets)
        out._polygons = polygons
        return out

    @classmethod
    def concatenate(cls, soups: Sequence['PolygonSoup']):
        soups = [cls.from_polygons(s) for s in soups]
        sizes = [0]
        for s in soups:
            sizes.append(sizes[-1] + s.smooth_xy.shape[0])
        offsets = np.concatenate([np.zeros(1, np.int64)] + [s.offsets[1:] + base for s, base in zip(soups, sizes)])
        smooth = np.concatenate([s.smooth_xy for s in soups], axis=0) if soups else np.zeros((0, 2), np.float64)
        return cls(smooth, offsets)

    @classmethod
    def from_np_arrays_dropping_closing_duplicates(cls, xy: np.ndarray, offsets: np.ndarray):
        """``Polygon.from_np_array`` per polygon (element/polygon.py:59-64, point.py:203-210): the values become the smooth
        positions; a polygon of more than two points whose first and last integer positions agree loses its last."""
        xy = np.asarray(xy).reshape(-1, 2)
        smooth = xy.astype(np.float64)
        offsets = np.asarray(offsets, dtype=np.int64)
        ints = _rint_xy(smooth)
        begin, end = offsets[:-1], offsets[1:]
        sizes = end - begin
        closing = (sizes > 2) & (ints[begin] == ints[np.maximum(end - 1, 0)]).all(axis=1)
        if closing.any():
            keep = np.ones(smooth.shape[0], bool)
            keep[end[closing] - 1] = False
            smooth = smooth[keep]
            new_offsets = np.zeros_like(offsets)
            new_offsets[1:] = np.cumsum(sizes - closing)
            offsets = new_offsets
        return cls(smooth, offsets)

    # ---- arrays
    @property
    def int_xy(self) -> np.ndarray:
        if self._int_xy is None:
            self._int_xy = _rint_xy(self.smooth_xy)
        return self._int_xy

    @property
    def num_polygons(self):
        return self.offsets.shape[0] - 1

    # ---- sequence protocol
    def __len__(self):
        return self.offsets.shape[0] - 1

    def _materialised(self):
        if self._polygons is None:
            from .polygon import Polygon
            self._polygons = [Polygon.from_smooth_xy(self.smooth_xy[b:e])
                              for b, e in zip(self.offsets[:-1], self.offsets[1:])]
        return self._polygons

    def __getitem__(self, index):
        if isinstance(index, slice):
            start, stop, step = index.indices(len(self))
            if step != 1:
                return PolygonSoup.from_polygons(self._materialised()[index])
            stop = max(stop, start)
            lo, hi = self.offsets[start], self.offsets[stop]
            out = PolygonSoup(self.smooth_xy[lo:hi], self.offsets[start:stop + 1] - lo)
            if self._int_xy is not None:
                out._int_xy = self._int_xy[lo:hi]
            if self._polygons is not None:
                out._polygons = self._polygons[start:stop]
            return out
        return self._materialised()[index]

    def __iter__(self):
        return iter(self._materialised())

    def __repr__(self):
        return f'PolygonSoup({len(self)} polygons, {self.smooth_xy.shape[0]} vertices)'

    def reordered(self, order: Sequence[int]):
        """The polygons ``order[0], order[1], ...`` as a new soup."""
        order = np.asarray(order, dtype=np.int64)
        begin, end = self.offsets[order], self.offsets[order + 1]
        sizes = end - begin
        offsets = np.zeros(order.shape[0] + 1, np.int64)
        offsets[1:] = np.cumsum(sizes)
        index = np.repeat(begin - offsets[:-1], sizes) + np.arange(offsets[-1])
        return PolygonSoup(self.smooth_xy[index], offsets)

    # ---- the polygons' operators, vectorised
    def with_smooth_xy(self, smooth_xy: np.ndarray):
        return PolygonSoup(smooth_xy, self.offsets)

    def to_clipped_polygons(self, shapable_or_shape):
        shape = extract_shape_from_shapable_or_shape(shapable_or_shape)
        clipped = _clip_to_shape(self.smooth_xy, self.int_xy, shape)
        return self if clipped is self.smooth_xy else self.with_smooth_xy(clipped)

    def to_shifted_polygons(self, offset_y: int = 0, offset_x: int = 0):
        return self.with_smooth_xy(self.smooth_xy + np.array([offset_x, offset_y], np.float64))

    def to_conducted_resized_polygons(self, shapable_or_shape, resized_height: Optional[int] = None,
                                      resized_width: Optional[int] = None):
        sizes = generate_shape_and_resized_shape(shapable_or_shape, resized_height, resized_width)
        return self.with_smooth_xy(_resize_xy(self.smooth_xy, sizes))
