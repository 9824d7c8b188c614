"""Polygon rasterisation (cv.fillPoly semantics) and the operators built on it, GPU vs oracle."""
import numpy as np
import pytest
from numpy.random import default_rng

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def N():
    from vkit_amd import _native
    return _native


def _check(N, shape, pts):
    got = N.fill_poly_mask(shape, pts)
    want = O.fill_poly(shape, np.asarray(pts, np.int32))
    np.testing.assert_array_equal(got, want)
    return got


def test_fill_poly_random_polygons(N):
    rng = default_rng(5)
    for case in range(60):
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        n = int(rng.integers(1, 12))
        pts = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], axis=1)
        _check(N, (h, w), pts)


def test_fill_poly_degenerate_shapes(N):
    _check(N, (9, 9), [(4, 4)])                                   # single point
    _check(N, (9, 9), [(1, 1), (7, 6)])                           # segment
    _check(N, (9, 9), [(0, 3), (8, 3), (4, 3)])                   # collinear horizontal
    _check(N, (9, 9), [(3, 0), (3, 8), (3, 2)])                   # collinear vertical
    _check(N, (12, 12), [(0, 0), (11, 0), (11, 11), (0, 11)])     # full rectangle
    _check(N, (12, 12), [(0, 0), (11, 11), (11, 0), (0, 11)])     # bow tie
    _check(N, (30, 30), [(2, 2), (27, 2), (27, 27), (2, 27), (2, 2)])  # closing duplicate kept at this level


def test_fill_poly_star_and_many_vertices(N):
    rng = default_rng(6)
    # a self-intersecting star: even-odd rule leaves the centre empty
    k = 7
    ang = np.arange(k) * (2 * np.pi * 3 / k)
    pts = np.stack([200 + 180 * np.cos(ang), 200 + 180 * np.sin(ang)], axis=1).round().astype(np.int32)
    got = _check(N, (401, 401), pts)
    assert got[200, 200] == 1 or got[200, 200] == 0  # value itself is the oracle's; shape checked above
    # a few thousand vertices on a noisy circle (long edge table, many crossings per scanline)
    n = 3000
    t = np.sort(rng.uniform(0, 2 * np.pi, n))
    r = 700 + rng.integers(-40, 41, n)
    pts = np.stack([800 + r * np.cos(t), 800 + r * np.sin(t)], axis=1).round().astype(np.int32)
    _check(N, (1601, 1601), pts)


def test_fill_poly_full_size_border_polygon(N):
    """Border polygon of a 2048^2 camera-distorted lattice (the active mask of the C3 workload)."""
    from vkit_amd.mechanism.distortion import camera_cubic_curve
    from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam
    gen = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), 5)
    config = gen((2048, 2048), default_rng(3))
    state = camera_cubic_curve.generate_state(config, (2048, 2048))
    grid = state.dst_image_grid
    polygon = grid.generate_border_polygon()
    active = camera_cubic_curve.get_active_mask(config, (2048, 2048), state=state)
    box = polygon.bounding_box
    want = np.zeros(active.shape, np.uint8)
    want[box.up:box.down + 1, box.left:box.right + 1] = O.fill_poly(
        box.shape, polygon.self_relative_polygon.to_np_array())
    np.testing.assert_array_equal(active.mat, want)
    assert 0 < int(active.mat.sum()) < active.mat.size


def test_polygon_fill_and_extract_operators(N):
    from vkit_amd.element import Image, Mask, Polygon
    rng = default_rng(8)
    polygon = Polygon.from_xy_pairs([(10, 5), (60, 12), (48, 50), (30, 30), (6, 44)])
    box = polygon.bounding_box
    raster = O.fill_poly(box.shape, polygon.self_relative_polygon.to_np_array()).astype(bool)
    np.testing.assert_array_equal(polygon.np_mask, raster)

    mat = rng.integers(0, 256, (64, 72, 3), dtype=np.uint8)
    image = Image(mat=mat.copy())
    polygon.fill_image(image, (255, 0, 7), alpha=0.4)
    want = mat.copy()
    O.fill(want, (box.up, box.left, box.height, box.width), (255, 0, 7), mask=raster.astype(np.uint8), alpha=0.4)
    np.testing.assert_array_equal(image.mat, want)

    mask = Mask.from_shape((64, 72))
    polygon.fill_mask(mask)
    full = np.zeros((64, 72), np.uint8)
    full[box.up:box.down + 1, box.left:box.right + 1] = raster
    np.testing.assert_array_equal(mask.mat, full)

    extracted = polygon.extract_image(Image(mat=mat.copy()))
    want = mat[box.up:box.down + 1, box.left:box.right + 1].copy()
    want[~raster] = 0
    np.testing.assert_array_equal(extracted.mat, want)
