"""Image-grid based geometric distortions (reference: grid_rendering/interface.py).

``DistortionStateImageGridBased`` owns a source / destination ``ImageGrid`` pair; ``FuncImageGridBased`` blends
Image / Mask / ScoreMap through ONE device-side grid (cell homographies + exact cv.fillPoly ownership + bilinear
gather, csrc/grid.hip).  Point and polygon LISTS go through the lattice in one device launch
(``vkx_grid_project_points``); a single ``func_point`` call solves its cell on the host."""
from typing import Generic, Optional, Tuple, Type, TypeVar

import numpy as np
from numpy.random import Generator as RandomGenerator

from vkit_amd import _native
from vkit_amd.element import Image, Mask, Point, PointArray, PointTuple, Polygon, PolygonSoup, ScoreMap
from ...interface import Distortion, DistortionConfig, DistortionState
from .grid_creator import create_dst_image_grid_and_shift_amounts_and_resize_ratios
from .image_grid import ImageGrid
from .point_projector import PointProjector

_T_CONFIG = TypeVar('_T_CONFIG', bound=DistortionConfig)


class DistortionStateImageGridBased(DistortionState[_T_CONFIG]):
    src_image_grid: ImageGrid
    dst_image_grid: ImageGrid
    shift_amount_y: float
    shift_amount_x: float
    resize_ratio_y: float
    resize_ratio_x: float

    def initialize_image_grid_based(self, src_image_grid: ImageGrid, point_projector: PointProjector,
                                    resize_as_src: bool = False):
        self.src_image_grid = src_image_grid
        (
            self.dst_image_grid,
            (self.shift_amount_y, self.shift_amount_x),
            (self.resize_ratio_y, self.resize_ratio_x),
        ) = create_dst_image_grid_and_shift_amounts_and_resize_ratios(src_image_grid, point_projector,
                                                                      resize_as_src=resize_as_src)

    def shift_and_resize_point(self, point: Point):
        return Point.create(
            y=(point.smooth_y - self.shift_amount_y) * self.resize_ratio_y,
            x=(point.smooth_x - self.shift_amount_x) * self.resize_ratio_x,
        )

    @property
    def result_shape(self):
        return self.dst_image_grid.image_height, self.dst_image_grid.image_width


_T_STATE = TypeVar('_T_STATE', bound=DistortionStateImageGridBased)


def blend_src_to_dst(mat: np.ndarray, src_image_grid: ImageGrid, dst_image_grid: ImageGrid) -> np.ndarray:
    """cv.remap(mat, map_x, map_y, INTER_LINEAR) through the grid, without materialising the maps."""
    return _native.grid_remap([mat], src_image_grid.vertices, dst_image_grid.vertices, dst_image_grid.image_shape)[0]


class FuncImageGridBased(Generic[_T_CONFIG, _T_STATE]):

    @classmethod
    def func_image(cls, config, state, image: Image, rng: Optional[RandomGenerator]):
        assert state
        return Image(mat=blend_src_to_dst(image.arr, state.src_image_grid, state.dst_image_grid), mode=image.mode)

    @classmethod
    def func_score_map(cls, config, state, score_map: ScoreMap, rng: Optional[RandomGenerator]):
        assert state
        return ScoreMap(mat=blend_src_to_dst(score_map.arr, state.src_image_grid, state.dst_image_grid))

    @classmethod
    def func_mask(cls, config, state, mask: Mask, rng: Optional[RandomGenerator]):
        # bilinear on the 0/1 bytes, exactly like the reference (grid_blender.py:74-81): no nearest neighbour
        assert state
        return Mask(mat=blend_src_to_dst(mask.arr, state.src_image_grid, state.dst_image_grid))

    @classmethod
    def func_active_mask(cls, config, state, shape: Tuple[int, int], rng: Optional[RandomGenerator]):
        assert state
        # reference grid_rendering/interface.py:177-192: border polygon of the destination lattice, filled with 1
        dst_grid = state.dst_image_grid
        active_mask = Mask.from_shape((dst_grid.image_height, dst_grid.image_width))
        dst_grid.generate_border_polygon().fill_mask(active_mask)
        return active_mask

    @classmethod
    def func_point(cls, config, state, shape: Tuple[int, int], point: Point, rng: Optional[RandomGenerator]):
        assert state
        src_grid, dst_grid = state.src_image_grid, state.dst_image_grid
        assert src_grid.grid_size
        row = point.y // src_grid.grid_size
        col = point.x // src_grid.grid_size
        trans_mat = src_grid.get_trans_mat(row, col, dst_grid)
        tx, ty, t = np.matmul(trans_mat, (point.smooth_x, point.smooth_y, 1.0))
        return Point.create(y=float(ty / t), x=float(tx / t))


    @classmethod
    def _project_arrays(cls, state, int_xy: np.ndarray, smooth_xy: np.ndarray) -> np.ndarray:
        """All points of one call through the lattice in ONE device launch (the reference loops point by point,
        distortion/interface.py:638-661): int (n, 2) rounded positions pick the cell, float64 (n, 2) smooth positions
        are mapped; returns float64 (n, 2)."""
        src_grid, dst_grid = state.src_image_grid, state.dst_image_grid
        assert src_grid.grid_size
        if smooth_xy.shape[0] == 0:
            return smooth_xy
        return np.array(_native.project_points(src_grid.vertices, dst_grid.vertices, src_grid.grid_size, int_xy, smooth_xy))

    @classmethod
    def func_points(cls, config, state, shape: Tuple[int, int], points, rng: Optional[RandomGenerator]):
        """Returns a ``PointArray`` (a sequence of ``Point`` over one array; ``PointTuple`` semantics)."""
        assert state
        points = PointArray.from_points(points, tuple_like=True)
        return PointArray(cls._project_arrays(state, points.int_xy, points.smooth_xy), tuple_like=True)

    @classmethod
    def func_polygons(cls, config, state, shape: Tuple[int, int], polygons, rng: Optional[RandomGenerator]):
        """Returns a ``PolygonSoup`` (a sequence of ``Polygon`` over one vertex array)."""
        assert state
        soup = PolygonSoup.from_polygons(polygons)
        return soup.with_smooth_xy(cls._project_arrays(state, soup.int_xy, soup.smooth_xy))


class DistortionImageGridBased(Distortion[_T_CONFIG, _T_STATE]):

    def __init__(self, config_cls: Type[_T_CONFIG], state_cls: Type[_T_STATE]):
        funcs = FuncImageGridBased[_T_CONFIG, _T_STATE]
        super().__init__(
            config_cls=config_cls,
            state_cls=state_cls,
            func_image=funcs.func_image,
            func_mask=funcs.func_mask,
            func_score_map=funcs.func_score_map,
            func_active_mask=funcs.func_active_mask,
            func_point=funcs.func_point,
            func_points=funcs.func_points,
            func_polygons=funcs.func_polygons,
        )

    def distort(self, config_or_config_generator, shapable_or_shape=None, image=None, mask=None, score_map=None,
                point=None, points=None, corner_points=None, polygon=None, polygons=None, get_active_mask=False,
                get_config=False, get_state=False, disable_clip_result_elements=False, rng=None):
        """Same signature and contract as ``Distortion.distort`` (distortion/interface.py:824-912); Image + Mask +
        ScoreMap of one call share ONE device pass (one ownership raster, one homography evaluation per pixel, three
        gathers) instead of three."""
        shared = [e for e in (image, mask, score_map) if e is not None]
        if len(shared) < 2:
            return super().distort(config_or_config_generator, shapable_or_shape, image, mask, score_map, point, points,
                                   corner_points, polygon, polygons, get_active_mask, get_config, get_state,
                                   disable_clip_result_elements, rng)
        if shapable_or_shape is None:
            shapable_or_shape = shared[0]
        # everything but the three pixel elements through the base operator (it also prepares config, rng and state)
        result = super().distort(config_or_config_generator, shapable_or_shape, None, None, None, point, points,
                                 corner_points, polygon, polygons, get_active_mask, get_config, True,
                                 disable_clip_result_elements, rng)
        state = result.state
        outs = _native.grid_remap([e.arr for e in shared], state.src_image_grid.vertices,
                                  state.dst_image_grid.vertices, state.dst_image_grid.image_shape)
        it = iter(outs)
        if image is not None:
            result.image = Image(mat=next(it), mode=image.mode)
            assert result.shape == result.image.shape
        if mask is not None:
            result.mask = Mask(mat=next(it))
            assert result.shape == result.mask.shape
        if score_map is not None:
            result.score_map = ScoreMap(mat=next(it))
            assert result.shape == result.score_map.shape
        if not get_state:
            result.state = None
        return result
