"""The array-native point / polygon containers (vkit_amd/element/soup.py) against the object-by-object operators of
element/point.py and element/polygon.py they stand in for (reference: vkit/element/point.py:57-120, polygon.py:59-64,
distortion/geometric/affine.py:65-82)."""
import numpy as np
import pytest

from vkit_amd.element import Point, PointArray, PointList, PointTuple, Polygon, PolygonSoup
from vkit_amd.mechanism.distortion.geometric import affine as A


def _random_points(rng, n, lo=-40.0, hi=140.0):
    xy = rng.uniform(lo, hi, (n, 2))
    whole = rng.random(n) < 0.2
    xy[whole] = np.round(xy[whole])                                                                      # some integers
    xy[::7] += 0.5 - (xy[::7] % 1.0)                                                                     # some exact halves
    return xy


def _same_points(array_points, object_points):
    assert len(array_points) == len(object_points)
    for a, b in zip(array_points, object_points):
        assert (a.smooth_x, a.smooth_y, a.x, a.y) == (b.smooth_x, b.smooth_y, b.x, b.y)


@pytest.mark.parametrize('seed', range(4))
def test_point_array_matches_point_list(seed):
    rng = np.random.default_rng(seed)
    xy = _random_points(rng, 257)
    objects = PointList(Point.create(y=float(y), x=float(x)) for x, y in xy)
    arr = PointArray(xy)
    _same_points(arr, objects)
    assert (arr.to_np_array() == objects.to_np_array()).all()
    assert (arr.to_smooth_np_array() == objects.to_smooth_np_array()).all()
    assert (arr.to_point_tuple().to_smooth_np_array() == objects.to_point_tuple().to_smooth_np_array()).all()
    shape = (100, 120)
    _same_points(arr.to_clipped_points(shape), objects.to_clipped_points(shape))
    _same_points(arr.to_shifted_points(offset_y=-7, offset_x=3), objects.to_shifted_points(offset_y=-7, offset_x=3))
    _same_points(arr.to_relative_points(5, 9), objects.to_relative_points(5, 9))
    clipped = arr.to_clipped_points(shape)
    _same_points(clipped.to_conducted_resized_points(shape, resized_height=77),
                 PointList(clipped).to_conducted_resized_points(shape, resized_height=77))
    _same_points(clipped.to_conducted_resized_points(shape, resized_height=131, resized_width=59),
                 PointList(clipped).to_conducted_resized_points(shape, resized_height=131, resized_width=59))
    assert arr == objects and arr[3:9] == objects[3:9]
    assert PointArray.from_points(objects).smooth_xy.tolist() == xy.tolist()


def _random_polygons(rng, n):
    polygons = []
    for _ in range(n):
        k = int(rng.integers(1, 7))
        xy = _random_points(rng, k, 0.0, 99.0)
        if k > 2 and rng.random() < 0.4:
            xy[-1] = xy[0] + rng.uniform(-0.3, 0.3, 2)         # a closing (near) duplicate
        polygons.append(Polygon.create(points=[Point.create(y=float(y), x=float(x)) for x, y in xy]))
    return polygons


@pytest.mark.parametrize('seed', range(4))
def test_polygon_soup_matches_polygons(seed):
    rng = np.random.default_rng(100 + seed)
    polygons = _random_polygons(rng, 61)
    soup = PolygonSoup.from_polygons(polygons)
    assert len(soup) == len(polygons)
    for a, b in zip(PolygonSoup(soup.smooth_xy, soup.offsets), polygons):
        _same_points(a.points, b.points)
    shape = (80, 90)
    for a, b in zip(soup.to_clipped_polygons(shape), polygons):
        _same_points(a.points, b.to_clipped_polygon(shape).points)
    for a, b in zip(soup.to_shifted_polygons(offset_y=4, offset_x=-11), polygons):
        _same_points(a.points, b.to_shifted_polygon(offset_y=4, offset_x=-11).points)
    clipped = soup.to_clipped_polygons(shape)
    for a, b in zip(clipped.to_conducted_resized_polygons(shape, resized_width=200), clipped):
        _same_points(a.points, b.to_conducted_resized_polygon(shape, resized_width=200).points)
    # slices, reordering, concatenation keep the polygons
    for a, b in zip(soup[5:17], polygons[5:17]):
        _same_points(a.points, b.points)
    order = rng.permutation(len(polygons))
    for a, idx in zip(soup.reordered(order), order):
        _same_points(a.points, polygons[idx].points)
    both = PolygonSoup.concatenate([soup[:10], polygons[10:20]])
    for a, b in zip(both, polygons[:20]):
        _same_points(a.points, b.points)
    # the integer vertex array the label paint consumes
    assert (soup.int_xy == np.concatenate([p.to_np_array() for p in polygons])).all()
    lazy = soup[0]
    assert (lazy.to_np_array() == polygons[0].to_np_array()).all() and lazy.bounding_box == polygons[0].bounding_box


def _reference_affine_polygons(trans_mat, polygons):
    # distortion/geometric/affine.py:70-82 of the reference, object by object
    flat, spans = PointList(), []
    for polygon in polygons:
        spans.append((len(flat), len(flat) + polygon.num_points))
        flat.extend(polygon.points)
    moved = A.affine_np_points(trans_mat, flat.to_smooth_np_array())
    return [Polygon.from_np_array(moved[b:e]) for b, e in spans]


@pytest.mark.parametrize('seed', range(3))
def test_affine_on_arrays_is_the_reference_construction(seed):
    rng = np.random.default_rng(200 + seed)
    polygons = _random_polygons(rng, 40)
    for trans_mat in (np.array([[0.8660254, -0.5, 12.0], [0.5, 0.8660254, 0.0]], np.float32),
                      np.array([[1.0, 0.2, 0.0], [0.0, 1.0, 0.0], [1e-3, 2e-3, 1.0]], np.float32),
                      np.array([[0.02, 0.0, 0.0], [0.0, 0.02, 0.0]], np.float32)):        # collapses vertices: closing duplicates
        want = _reference_affine_polygons(trans_mat, polygons)
        got = A.affine_polygons(trans_mat, polygons)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            _same_points(a.points, b.points)
        points = PointTuple(p for polygon in polygons[:9] for p in polygon.points)
        want_points = PointTuple.from_np_array(A.affine_np_points(trans_mat, points.to_smooth_np_array()))
        _same_points(A.affine_points(trans_mat, points), want_points)
        _same_points(A.affine_points(trans_mat, PointArray.from_points(points)), want_points)


def test_group_means_equal_the_slice_means():
    """``page_distortion._group_means``: the text-line heights of a page as whole-array arithmetic, bit for bit ``slice.mean()``
    (float32 and float64, groups of 1 .. 7 elements vectorised, larger groups through the slices)."""
    from numpy.random import default_rng
    from vkit_amd.pipeline.text_detection.page_distortion import _group_means
    rng = default_rng(0)
    for dtype in (np.float32, np.float64):
        for hi in (8, 40):
            for _ in range(200):
                sizes = [int(v) for v in rng.integers(1, hi, int(rng.integers(1, 80)))]
                values = (rng.random(sum(sizes)) * 100).astype(dtype)
                want, begin = [], 0
                for size in sizes:
                    want.append(float(values[begin:begin + size].mean()))
                    begin += size
                assert _group_means(values, sizes) == want
    assert _group_means(np.zeros(0), []) == []
