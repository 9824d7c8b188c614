"""The ``Distortion`` operator: the drop-in boundary of the accelerated path.

Same constructor, callable signatures, rng contract and element order as the reference operator
(vkit/mechanism/distortion/interface.py:138-912):

* ``Distortion(config_cls, state_cls, func_image, func_mask=None, func_score_map=None, func_active_mask=None,
  func_point=None, func_points=None, func_polygon=None, func_polygons=None)``;
* config: instance | mapping | ``generator(shape, rng)``; configs with ``supports_rng_state`` capture the caller's
  bit-generator state once, advance the caller's rng by one ``rng.random()`` and run on a private generator
  that is rewound before every element (reference :261-307, :132-135);
* ``state_cls(config, shape, rng)`` unless it is ``DistortionNopState[...]``;
* ``distort`` processes image, mask, score_map, point, points, corner_points, polygon, polygons, active_mask in
  that order and clips points / polygons of geometric results to the result shape.

The ``func_*`` callables registered by this package call HIP kernels through ``vkit_amd._native``.
"""
from typing import Any, Callable, Generic, Iterable, Mapping, Optional, Sequence, Tuple, Type, TypeVar, Union, get_origin

import attrs
from numpy.random import Generator as RandomGenerator
from numpy.random import default_rng

from vkit_amd.element import Image, Mask, Point, PointArray, PointList, PointTuple, Polygon, PolygonSoup, ScoreMap, Shapable
from vkit_amd.utility import dyn_structure, get_config_class_snake_case_name


class DistortionConfig:
    _cached_name: str = ''

    @classmethod
    def get_name(cls):
        if not cls._cached_name:
            cls._cached_name = get_config_class_snake_case_name(cls.__name__)
        return cls._cached_name

    @property
    def name(self):
        return self.get_name()

    @property
    def supports_rng_state(self) -> bool:
        return False

    @property
    def rng_state(self) -> Optional[Mapping[str, Any]]:
        return None

    @rng_state.setter
    def rng_state(self, val: Mapping[str, Any]):
        pass


_T_CONFIG = TypeVar('_T_CONFIG', bound=DistortionConfig)


class DistortionState(Generic[_T_CONFIG]):

    def __init__(self, config: _T_CONFIG, shape: Tuple[int, int], rng: Optional[RandomGenerator]):
        raise NotImplementedError()

    @property
    def result_shape(self) -> Optional[Tuple[int, int]]:
        return None


class DistortionNopState(DistortionState[_T_CONFIG]):
    """Marker for stateless distortions; never instantiated."""


_T_STATE = TypeVar('_T_STATE', bound=DistortionState)


@attrs.define
class DistortionResult:
    shape: Tuple[int, int]
    image: Optional[Image] = None
    mask: Optional[Mask] = None
    score_map: Optional[ScoreMap] = None
    active_mask: Optional[Mask] = None
    point: Optional[Point] = None
    points: Optional[PointTuple] = None
    corner_points: Optional[PointTuple] = None
    polygon: Optional[Polygon] = None
    polygons: Optional[Sequence[Polygon]] = None
    config: Optional[Any] = None
    state: Optional[Any] = None
    meta: Optional[Mapping[str, Any]] = None


@attrs.define
class DistortionInternals(Generic[_T_CONFIG, _T_STATE]):
    config: _T_CONFIG
    state: Optional[_T_STATE]
    shape: Tuple[int, int]
    rng: Optional[RandomGenerator]

    def restore_rng_if_supported(self):
        if self.rng:
            assert self.config.supports_rng_state and self.config.rng_state
            self.rng.bit_generator.state = self.config.rng_state


_ConfigLike = Union[DistortionConfig, Mapping[str, Any], Callable[[Tuple[int, int], RandomGenerator], Any]]


class Distortion(Generic[_T_CONFIG, _T_STATE]):

    def __init__(
        self,
        config_cls: Type[_T_CONFIG],
        state_cls: Type[_T_STATE],
        func_image: Callable[[_T_CONFIG, Optional[_T_STATE], Image, Optional[RandomGenerator]], Image],
        func_mask: Optional[Callable[..., Mask]] = None,
        func_score_map: Optional[Callable[..., ScoreMap]] = None,
        func_active_mask: Optional[Callable[..., Mask]] = None,
        func_point: Optional[Callable[..., Point]] = None,
        func_points: Optional[Callable[..., PointTuple]] = None,
        func_polygon: Optional[Callable[..., Polygon]] = None,
        func_polygons: Optional[Callable[..., Sequence[Polygon]]] = None,
    ):
        self.config_cls = config_cls
        self.state_cls = state_cls
        self.func_image = func_image
        self.func_score_map = func_score_map
        self.func_mask = func_mask
        self.func_active_mask = func_active_mask
        self.func_point = func_point
        self.func_points = func_points
        self.func_polygon = func_polygon
        self.func_polygons = func_polygons

    @property
    def is_geometric(self):
        return any((self.func_point, self.func_points, self.func_polygon, self.func_polygons, self.func_active_mask))

    # ------------------------------------------------------------------ config / rng / state preparation
    def prepare_config_and_rng(self, config_or_config_generator: _ConfigLike, shape: Tuple[int, int],
                               rng: Optional[RandomGenerator]):
        if callable(config_or_config_generator):
            if not rng:
                raise RuntimeError('config_generator but rng is None.')
            raw = config_or_config_generator(shape, rng)
        else:
            raw = config_or_config_generator
        config = dyn_structure(raw, self.config_cls)

        if not config.supports_rng_state:
            return config, None  # rng is deliberately withheld from rng-free distortions

        if not config.rng_state:
            if not rng:
                raise RuntimeError('both config.rng_state and rng are None.')
            config.rng_state = rng.bit_generator.state
            rng.random()  # move the caller's stream on so the next distortion differs
        private_rng = default_rng()
        private_rng.bit_generator.state = config.rng_state
        return config, private_rng

    @classmethod
    def get_shape_from_shapable_or_shape(cls, shapable_or_shape: Union[Shapable, Tuple[int, int]]):
        if isinstance(shapable_or_shape, (list, tuple)):
            assert len(shapable_or_shape) == 2
            return shapable_or_shape
        return shapable_or_shape.shape

    def prepare_internals(self, config_or_config_generator: _ConfigLike, state: Optional[_T_STATE],
                          shapable_or_shape: Union[Shapable, Tuple[int, int]], rng: Optional[RandomGenerator] = None,
                          disable_state_initialization: bool = False):
        shape = self.get_shape_from_shapable_or_shape(shapable_or_shape)
        config, rng = self.prepare_config_and_rng(config_or_config_generator, shape, rng)
        if get_origin(self.state_cls) is DistortionNopState:
            state = None
        elif state is None and not disable_state_initialization:
            state = self.state_cls(config, shape, rng)
        return DistortionInternals(config, state, shape, rng)

    def generate_config_and_state(self, config_or_config_generator: _ConfigLike, state: Optional[_T_STATE],
                                  shapable_or_shape, rng: Optional[RandomGenerator] = None):
        internals = self.prepare_internals(config_or_config_generator, state, shapable_or_shape, rng)
        return internals.config, internals.state

    def generate_config(self, config_or_config_generator: _ConfigLike, shapable_or_shape,
                        rng: Optional[RandomGenerator] = None):
        return self.prepare_internals(config_or_config_generator, None, shapable_or_shape, rng,
                                      disable_state_initialization=True).config

    def generate_state(self, config_or_config_generator: _ConfigLike, shapable_or_shape,
                       rng: Optional[RandomGenerator] = None):
        return self.prepare_internals(config_or_config_generator, None, shapable_or_shape, rng).state

    # ------------------------------------------------------------------ per-element application
    def distort_image_based_on_internals(self, internals: DistortionInternals, image: Image):
        internals.restore_rng_if_supported()
        return self.func_image(internals.config, internals.state, image, internals.rng)

    def distort_score_map_based_on_internals(self, internals: DistortionInternals, score_map: ScoreMap):
        internals.restore_rng_if_supported()
        if not self.func_score_map:
            return score_map  # photometric distortions leave score maps alone (same object)
        return self.func_score_map(internals.config, internals.state, score_map, internals.rng)

    def distort_mask_based_on_internals(self, internals: DistortionInternals, mask: Mask):
        internals.restore_rng_if_supported()
        if not self.func_mask:
            return mask
        return self.func_mask(internals.config, internals.state, mask, internals.rng)

    def get_active_mask_based_on_internals(self, internals: DistortionInternals):
        internals.restore_rng_if_supported()
        if self.func_active_mask:
            return self.func_active_mask(internals.config, internals.state, internals.shape, internals.rng)
        return self.distort_mask_based_on_internals(internals, Mask.from_shape(internals.shape, value=1))

    def distort_point_based_on_internals(self, internals: DistortionInternals, point: Point):
        internals.restore_rng_if_supported()
        if self.func_point:
            return self.func_point(internals.config, internals.state, internals.shape, point, internals.rng)
        if self.func_points:
            return self.func_points(internals.config, internals.state, internals.shape, [point], internals.rng)[0]
        if self.is_geometric:
            raise RuntimeError('Missing self.func_points or self.func_point.')
        return point

    def distort_points_based_on_internals(self, internals: DistortionInternals,
                                          points: Union[PointList, PointTuple, Iterable[Point]]):
        internals.restore_rng_if_supported()
        if isinstance(points, PointArray):
            # array-native container (element/soup.py): stays an array through photometric operators and through the
            # batched geometric ones
            if not self.is_geometric:
                return points.to_point_tuple()
            if self.func_points:
                return self.func_points(internals.config, internals.state, internals.shape, points, internals.rng)
        points = PointList(points)
        if not self.is_geometric:
            # photometric distortions move nothing: the reference walks the points one by one to return each unchanged
            return points.to_point_tuple()
        if self.func_points:
            moved = self.func_points(internals.config, internals.state, internals.shape, points, internals.rng)
            # the batched operators answer with a PointArray; a caller that handed in point objects gets the
            # reference's container back
            return PointTuple(moved) if isinstance(moved, PointArray) else moved
        return PointList(self.distort_point_based_on_internals(internals, point) for point in points).to_point_tuple()

    def distort_polygon_based_on_internals(self, internals: DistortionInternals, polygon: Polygon):
        internals.restore_rng_if_supported()
        if self.func_polygon:
            return self.func_polygon(internals.config, internals.state, internals.shape, polygon, internals.rng)
        if self.func_polygons:
            return self.func_polygons(internals.config, internals.state, internals.shape, [polygon], internals.rng)[0]
        return Polygon.create(points=self.distort_points_based_on_internals(internals, polygon.points))

    def distort_polygons_based_on_internals(self, internals: DistortionInternals, polygons: Iterable[Polygon]):
        internals.restore_rng_if_supported()
        if not self.is_geometric:
            # unchanged (the reference rebuilds equal polygons point by point); a PolygonSoup stays one
            return polygons if isinstance(polygons, PolygonSoup) else list(polygons)
        if self.func_polygons:
            moved = self.func_polygons(internals.config, internals.state, internals.shape, polygons, internals.rng)
            if isinstance(moved, PolygonSoup) and not isinstance(polygons, PolygonSoup):
                return list(moved)         # polygon objects in, polygon objects out
            return moved
        return [self.distort_polygon_based_on_internals(internals, polygon) for polygon in polygons]

    # ------------------------------------------------------------------ single-element conveniences
    def distort_image(self, config_or_config_generator: _ConfigLike, image: Image, state: Optional[_T_STATE] = None,
                      rng: Optional[RandomGenerator] = None):
        internals = self.prepare_internals(config_or_config_generator, state, image, rng)
        return self.distort_image_based_on_internals(internals, image)

    def distort_score_map(self, config_or_config_generator: _ConfigLike, score_map: ScoreMap,
                          state: Optional[_T_STATE] = None, rng: Optional[RandomGenerator] = None):
        internals = self.prepare_internals(config_or_config_generator, state, score_map, rng)
        return self.distort_score_map_based_on_internals(internals, score_map)

    def distort_mask(self, config_or_config_generator: _ConfigLike, mask: Mask, state: Optional[_T_STATE] = None,
                     rng: Optional[RandomGenerator] = None):
        internals = self.prepare_internals(config_or_config_generator, state, mask, rng)
        return self.distort_mask_based_on_internals(internals, mask)

    def get_active_mask(self, config_or_config_generator: _ConfigLike, shapable_or_shape,
                        state: Optional[_T_STATE] = None, rng: Optional[RandomGenerator] = None):
        internals = self.prepare_internals(config_or_config_generator, state, shapable_or_shape, rng)
        return self.get_active_mask_based_on_internals(internals)

    def distort_point(self, config_or_config_generator: _ConfigLike, shapable_or_shape, point: Point,
                      state: Optional[_T_STATE] = None, rng: Optional[RandomGenerator] = None):
        internals = self.prepare_internals(config_or_config_generator, state, shapable_or_shape, rng)
        return self.distort_point_based_on_internals(internals, point)

    def distort_points(self, config_or_config_generator: _ConfigLike, shapable_or_shape, points,
                       state: Optional[_T_STATE] = None, rng: Optional[RandomGenerator] = None):
        internals = self.prepare_internals(config_or_config_generator, state, shapable_or_shape, rng)
        return self.distort_points_based_on_internals(internals, points)

    def distort_polygon(self, config_or_config_generator: _ConfigLike, shapable_or_shape, polygon: Polygon,
                        state: Optional[_T_STATE] = None, rng: Optional[RandomGenerator] = None):
        internals = self.prepare_internals(config_or_config_generator, state, shapable_or_shape, rng)
        return self.distort_polygon_based_on_internals(internals, polygon)

    def distort_polygons(self, config_or_config_generator: _ConfigLike, shapable_or_shape, polygons,
                         state: Optional[_T_STATE] = None, rng: Optional[RandomGenerator] = None):
        internals = self.prepare_internals(config_or_config_generator, state, shapable_or_shape, rng)
        return self.distort_polygons_based_on_internals(internals, polygons)

    # ------------------------------------------------------------------ the operator entry
    @classmethod
    def get_shape(cls, shapable_or_shape=None, image: Optional[Image] = None, mask: Optional[Mask] = None,
                  score_map: Optional[ScoreMap] = None):
        if shapable_or_shape is None:
            shapable_or_shape = image or mask or score_map
        assert shapable_or_shape
        return cls.get_shape_from_shapable_or_shape(shapable_or_shape)

    def clip_result_elements(self, result: DistortionResult):
        if not self.is_geometric:
            return
        if result.point:
            result.point = result.point.to_clipped_point(result.shape)
        if result.points:
            result.points = result.points.to_clipped_points(result.shape)
        if result.corner_points:
            result.corner_points = result.corner_points.to_clipped_points(result.shape)
        if result.polygon:
            result.polygon = result.polygon.to_clipped_polygon(result.shape)
        if result.polygons:
            if isinstance(result.polygons, PolygonSoup):
                result.polygons = result.polygons.to_clipped_polygons(result.shape)
            else:
                result.polygons = [polygon.to_clipped_polygon(result.shape) for polygon in result.polygons]

    def distort(
        self,
        config_or_config_generator: _ConfigLike,
        shapable_or_shape: Optional[Union[Shapable, Tuple[int, int]]] = None,
        image: Optional[Image] = None,
        mask: Optional[Mask] = None,
        score_map: Optional[ScoreMap] = None,
        point: Optional[Point] = None,
        points: Optional[Union[PointList, PointTuple, Iterable[Point]]] = None,
        corner_points: Optional[Union[PointList, PointTuple, Iterable[Point]]] = None,
        polygon: Optional[Polygon] = None,
        polygons: Optional[Iterable[Polygon]] = None,
        get_active_mask: bool = False,
        get_config: bool = False,
        get_state: bool = False,
        disable_clip_result_elements: bool = False,
        rng: Optional[RandomGenerator] = None,
    ):
        # names an EARLIER call left behind (it raised after an out-of-path operator had passed its image through) are not this call's
        from vkit_amd.mechanism.distortion.photometric.opt import take_passed_through
        take_passed_through()
        shape = self.get_shape(shapable_or_shape=shapable_or_shape, image=image, mask=mask, score_map=score_map)
        internals = self.prepare_internals(config_or_config_generator, None, shape, rng)

        result = DistortionResult(shape=shape)
        if self.is_geometric:
            assert internals.state and internals.state.result_shape
            result.shape = internals.state.result_shape

        if image:
            result.image = self.distort_image_based_on_internals(internals, image)
            assert result.shape == result.image.shape
        if mask:
            result.mask = self.distort_mask_based_on_internals(internals, mask)
            assert result.shape == result.mask.shape
        if score_map:
            result.score_map = self.distort_score_map_based_on_internals(internals, score_map)
            assert result.shape == result.score_map.shape
        if point:
            result.point = self.distort_point_based_on_internals(internals, point)
        if points:
            result.points = self.distort_points_based_on_internals(internals, points)
        if corner_points:
            result.corner_points = self.distort_points_based_on_internals(internals, corner_points)
        if polygon:
            result.polygon = self.distort_polygon_based_on_internals(internals, polygon)
        if polygons:
            result.polygons = self.distort_polygons_based_on_internals(internals, polygons)
        if get_active_mask:
            result.active_mask = self.get_active_mask_based_on_internals(internals)
            assert result.shape == result.active_mask.shape
        if get_config:
            result.config = internals.config
        if get_state:
            result.state = internals.state
        if not disable_clip_result_elements:
            self.clip_result_elements(result)
        # an operator outside the accelerated path that handed its image through says so in the result (photometric/opt.py)
        passed = take_passed_through()
        if passed:
            result.meta = dict(result.meta or {}, out_of_path=tuple(passed))
        return result
