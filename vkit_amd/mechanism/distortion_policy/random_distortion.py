"""``RandomDistortion``: staged random application of distortion policies (reference:
distortion_policy/random_distortion.py).

Stage 0 draws 0..2 photometric policies, stage 1 (prob 0.75) exactly one geometric policy, optionally followed
by a forced rotate stage.  The policy TABLE (names, order, weights) is the reference's, so that a given rng
state selects the same policies.  One member (``UNSUPPORTED_POLICY_NAMES``: ``jpeg_quality``) samples its config like the
reference but leaves the image unchanged: its pixel work is outside the accelerated path.
"""
import logging
from collections import defaultdict
from typing import Any, Iterable, List, Mapping, Optional, Sequence, Tuple, Union

import attrs
from numpy.random import Generator as RandomGenerator

from vkit_amd.element import Box, Image, Mask, Point, PointArray, PointList, PointTuple, Polygon, PolygonSoup, ScoreMap, Shapable
from vkit_amd.utility import PathType, dyn_structure, normalize_to_probs, rng_choice_with_size
from ..distortion.interface import Distortion, DistortionResult
from .geometric import affine, camera, mls
from .opt import LEVEL_MAX, LEVEL_MIN
from .photometric import blur, color, effect, noise, streak
from .type import DistortionPolicy

logger = logging.getLogger(__name__)


# Policies whose operator passes the image through unchanged (the pixel work is outside this path, see
# distortion/photometric/opt.py: pass_through_out_of_path); their configs are sampled like the reference's, so the
# table, the sampling and the rng stream are the reference's whether or not they are listed in disabled_policy_names.
UNSUPPORTED_POLICY_NAMES = (
    'jpeg_quality',
)


@attrs.define
class RandomDistortionDebug:
    distortion_names: List[str] = attrs.field(factory=list)
    distortion_levels: List[int] = attrs.field(factory=list)
    distortion_images: List[Image] = attrs.field(factory=list)
    distortion_configs: List[Any] = attrs.field(factory=list)
    distortion_states: List[Any] = attrs.field(factory=list)


@attrs.define
class RandomDistortionStageConfig:
    distortion_policies: Sequence[DistortionPolicy]
    distortion_policy_weights: Sequence[float]
    prob_enable: float
    num_distortions_min: int
    num_distortions_max: int
    inject_corner_points: bool = False
    conflict_control_keyword_groups: Sequence[Sequence[str]] = ()
    force_sample_level_in_full_range: bool = False


def _border_points(height: int, width: int):
    """Points on the four image edges, a quarter of the short side apart, corners included once."""
    step = min(height // 4, width // 4)
    assert step > 0
    ys = list(range(0, height, step))
    if ys[-1] < height - 1:
        ys.append(height - 1)
    xs = list(range(0, width, step))
    if xs[0] == 0:
        xs.pop(0)
    if xs[-1] == width - 1:
        xs.pop()
    points = PointList()
    for x in (0, width - 1):
        points.extend(Point.create(y=y, x=x) for y in ys)
    for y in (0, height - 1):
        points.extend(Point.create(y=y, x=x) for x in xs)
    return points.to_point_tuple()


class RandomDistortionStage:

    def __init__(self, config: RandomDistortionStageConfig):
        self.config = config
        self.distortion_policy_probs = normalize_to_probs(self.config.distortion_policy_weights)

    def _conflicts(self, policies) -> bool:
        """True if two sampled policies fall into the same keyword group (e.g. two blurs, two noises)."""
        hits = defaultdict(int)
        for policy in policies:
            for group_idx, keywords in enumerate(self.config.conflict_control_keyword_groups):
                if any(keyword in policy.name for keyword in keywords):
                    hits[group_idx] += 1
                    break
        return any(count > 1 for count in hits.values())

    def sample_distortion_policies(self, rng: RandomGenerator) -> Sequence[DistortionPolicy]:
        num_distortions = rng.integers(self.config.num_distortions_min, self.config.num_distortions_max + 1)
        if num_distortions <= 0:
            return ()
        for _ in range(5):
            policies = rng_choice_with_size(rng, self.config.distortion_policies, size=num_distortions,
                                            probs=self.distortion_policy_probs, replace=False)
            if not self._conflicts(policies):
                return policies
            logger.debug('distortion policies conflict detected, resampling')
        logger.warning(f'Cannot sample distortion policies with num_distortion={num_distortions}.')
        return ()

    def apply_distortions(self, distortion_result: DistortionResult, level_min: int, level_max: int,
                          rng: RandomGenerator, debug: Optional[RandomDistortionDebug] = None):
        if rng.random() > self.config.prob_enable:
            return distortion_result
        if self.config.inject_corner_points:
            distortion_result.corner_points = _border_points(*distortion_result.shape)
        if self.config.force_sample_level_in_full_range:
            level_min, level_max = LEVEL_MIN, LEVEL_MAX

        for policy in self.sample_distortion_policies(rng):
            level = rng.integers(level_min, level_max + 1)
            passed = tuple((distortion_result.meta or {}).get('out_of_path', ()))
            distortion_result = policy.distort(
                level=level,
                shapable_or_shape=distortion_result.shape,
                image=distortion_result.image,
                mask=distortion_result.mask,
                score_map=distortion_result.score_map,
                point=distortion_result.point,
                points=distortion_result.points,
                corner_points=distortion_result.corner_points,
                polygon=distortion_result.polygon,
                polygons=distortion_result.polygons,
                rng=rng,
                enable_debug=bool(debug),
            )
            if debug:
                assert distortion_result.image
                debug.distortion_images.append(distortion_result.image)
                debug.distortion_names.append(policy.name)
                debug.distortion_levels.append(level)
                debug.distortion_configs.append(distortion_result.config)
                debug.distortion_states.append(distortion_result.state)
            distortion_result.config = None
            distortion_result.state = None
            passed += tuple((distortion_result.meta or {}).get('out_of_path', ()))
            if passed:      # every operator of the chain that handed the image through unchanged (photometric/opt.py)
                distortion_result.meta = dict(distortion_result.meta or {}, out_of_path=passed)
        return distortion_result


class RandomDistortion:

    def __init__(self, configs: Sequence[RandomDistortionStageConfig], level_min: int, level_max: int,
                 out_of_path: Optional[str] = None):
        """``out_of_path``: what the operators outside the accelerated path (``UNSUPPORTED_POLICY_NAMES``) do with the image when
        they are drawn -- 'pass_through' (config sampled, image unchanged, the result's ``meta['out_of_path']`` names them) or
        'raise'; None leaves it to the surrounding ``out_of_path(...)`` context / the environment (photometric/opt.py)."""
        self.stages = [RandomDistortionStage(config) for config in configs]
        self.level_min = level_min
        self.level_max = level_max
        self.out_of_path = out_of_path

    @classmethod
    def get_distortion_result_all_points(cls, distortion_result: DistortionResult):
        if distortion_result.corner_points:
            yield from distortion_result.corner_points
        if distortion_result.point:
            yield distortion_result.point
        if distortion_result.points:
            yield from distortion_result.points
        if distortion_result.polygon:
            yield from distortion_result.polygon.points
        if distortion_result.polygons:
            for polygon in distortion_result.polygons:
                yield from polygon.points

    @classmethod
    def get_distortion_result_element_bounding_box(cls, distortion_result: DistortionResult):
        assert distortion_result.corner_points
        points = list(cls.get_distortion_result_all_points(distortion_result))
        return Box(up=min(p.y for p in points), down=max(p.y for p in points), left=min(p.x for p in points),
                   right=max(p.x for p in points))

    @classmethod
    def trim_distortion_result(cls, distortion_result: DistortionResult):
        """Crops the result to the bounding box of all tracked points (only when corner points were injected)."""
        if not distortion_result.corner_points:
            return distortion_result
        height, width = distortion_result.shape
        box = cls.get_distortion_result_element_bounding_box(distortion_result)
        pad_up, pad_down = box.up, height - 1 - box.down
        pad_left, pad_right = box.left, width - 1 - box.right
        assert min(pad_up, pad_down, pad_left, pad_right) >= -1  # rounding slack
        if max(pad_up, pad_down, pad_left, pad_right) <= 0:
            return distortion_result

        window = dict(up=max(0, box.up), down=min(height - 1, box.down), left=max(0, box.left),
                      right=min(width - 1, box.right))
        shift = dict(offset_y=-max(0, pad_up), offset_x=-max(0, pad_left))
        if distortion_result.image:
            distortion_result.image = distortion_result.image.to_cropped_image(**window)
        if distortion_result.mask:
            distortion_result.mask = distortion_result.mask.to_cropped_mask(**window)
        if distortion_result.score_map:
            distortion_result.score_map = distortion_result.score_map.to_cropped_score_map(**window)
        if distortion_result.point:
            distortion_result.point = distortion_result.point.to_shifted_point(**shift)
        if distortion_result.points:
            distortion_result.points = distortion_result.points.to_shifted_points(**shift)
        if distortion_result.polygon:
            distortion_result.polygon = distortion_result.polygon.to_shifted_polygon(**shift)
        if distortion_result.polygons:
            if isinstance(distortion_result.polygons, PolygonSoup):
                distortion_result.polygons = distortion_result.polygons.to_shifted_polygons(**shift)
            else:
                distortion_result.polygons = [p.to_shifted_polygon(**shift) for p in distortion_result.polygons]
        return distortion_result

    def distort(
        self,
        rng: RandomGenerator,
        shapable_or_shape: Optional[Union[Shapable, Tuple[int, int]]] = None,
        image: Optional[Image] = None,
        mask: Optional[Mask] = None,
        score_map: Optional[ScoreMap] = None,
        point: Optional[Point] = None,
        points: Optional[Union[PointList, PointTuple, Iterable[Point]]] = None,
        polygon: Optional[Polygon] = None,
        polygons: Optional[Iterable[Polygon]] = None,
        debug: Optional[RandomDistortionDebug] = None,
    ):
        shape = Distortion.get_shape(shapable_or_shape=shapable_or_shape, image=image, mask=mask,
                                     score_map=score_map)
        # array-native containers (element/soup.py) pass through as they are
        if points and not isinstance(points, PointArray):
            points = PointTuple(points)
        result = DistortionResult(shape=shape, image=image, mask=mask, score_map=score_map, point=point,
                                  points=(points.to_point_tuple() if isinstance(points, PointArray) else points) if points else None,
                                  polygon=polygon)
        if polygons:
            result.polygons = polygons if isinstance(polygons, PolygonSoup) else tuple(polygons)
        # precedence: this object's own setting (when given) over a surrounding ``with out_of_path(...)`` of the caller, over
        # the environment (tests/test_host_golden.py pins it)
        from vkit_amd.mechanism.distortion.photometric.opt import out_of_path
        with out_of_path(self.out_of_path):
            for stage in self.stages:
                result = stage.apply_distortions(result, self.level_min, self.level_max, rng, debug=debug)
        return self.trim_distortion_result(result)


@attrs.define
class RandomDistortionFactoryConfig:
    # photometric stage
    prob_photometric: float = 1.0
    num_photometric_min: int = 0
    num_photometric_max: int = 2
    photometric_conflict_control_keyword_groups: Sequence[Sequence[str]] = attrs.field(
        factory=lambda: [['blur', 'pixelation', 'jpeg'], ['noise']])
    # geometric stage
    prob_geometric: float = 0.75
    force_post_rotate: bool = False
    # shared
    level_min: int = LEVEL_MIN
    level_max: int = LEVEL_MAX
    disabled_policy_names: Sequence[str] = attrs.field(factory=list)
    name_to_policy_config: Mapping[str, Any] = attrs.field(factory=dict)
    name_to_policy_weight: Mapping[str, float] = attrs.field(factory=dict)


# (policy factories of one family, summed weight of the family); order is the reference's.
_PHOTOMETRIC_POLICY_FACTORIES_AND_DEFAULT_WEIGHTS_SUM_PAIRS = (
    ((color.mean_shift_policy_factory, color.color_shift_policy_factory, color.brightness_shift_policy_factory, color.std_shift_policy_factory,
      color.boundary_equalization_policy_factory, color.histogram_equalization_policy_factory, color.complement_policy_factory,
      color.posterization_policy_factory, color.color_balance_policy_factory, color.channel_permutation_policy_factory), 10.0),
    ((blur.gaussian_blur_policy_factory, blur.defocus_blur_policy_factory, blur.motion_blur_policy_factory,
      blur.glass_blur_policy_factory,
      blur.zoom_in_blur_policy_factory), 1.0),
    ((noise.gaussion_noise_policy_factory, noise.poisson_noise_policy_factory, noise.impulse_noise_policy_factory,
      noise.speckle_noise_policy_factory), 3.0),
    ((effect.jpeg_quality_policy_factory, effect.pixelation_policy_factory, effect.fog_policy_factory), 1.0),
    ((streak.line_streak_policy_factory, streak.rectangle_streak_policy_factory,
      streak.ellipse_streak_policy_factory), 1.0),
)

_GEOMETRIC_POLICY_FACTORIES_AND_DEFAULT_WEIGHTS_SUM_PAIRS = (
    ((affine.shear_hori_policy_factory, affine.shear_vert_policy_factory, affine.rotate_policy_factory,
      affine.skew_hori_policy_factory, affine.skew_vert_policy_factory), 1.0),
    ((mls.similarity_mls_policy_factory,), 1.0),
    ((camera.camera_plane_only_policy_factory, camera.camera_cubic_curve_policy_factory,
      camera.camera_plane_line_fold_policy_factory, camera.camera_plane_line_curve_policy_factory), 1.0),
)


class RandomDistortionFactory:

    @classmethod
    def unpack_policy_factories_and_default_weights_sum_pairs(cls, pairs):
        factories, weights = [], []
        for family, weights_sum in pairs:
            factories.extend(family)
            weights.extend([weights_sum / len(family)] * len(family))
        return factories, weights

    def __init__(self, photometric_policy_factories_and_default_weights_sum_pairs=
                 _PHOTOMETRIC_POLICY_FACTORIES_AND_DEFAULT_WEIGHTS_SUM_PAIRS,
                 geometric_policy_factories_and_default_weights_sum_pairs=
                 _GEOMETRIC_POLICY_FACTORIES_AND_DEFAULT_WEIGHTS_SUM_PAIRS):
        self.photometric_policy_factories, self.photometric_policy_default_weights = \
            self.unpack_policy_factories_and_default_weights_sum_pairs(
                photometric_policy_factories_and_default_weights_sum_pairs)
        self.geometric_policy_factories, self.geometric_policy_default_weights = \
            self.unpack_policy_factories_and_default_weights_sum_pairs(
                geometric_policy_factories_and_default_weights_sum_pairs)

    @classmethod
    def create_policies_and_policy_weights(cls, policy_factories, policy_default_weights,
                                           config: RandomDistortionFactoryConfig):
        disabled = set(config.disabled_policy_names)
        policies, weights = [], []
        for factory, default_weight in zip(policy_factories, policy_default_weights):
            if factory.name in disabled:
                continue
            policies.append(factory.create(config.name_to_policy_config.get(factory.name)))
            weights.append(config.name_to_policy_weight.get(factory.name, default_weight))
        return policies, weights

    def create(self, config: Optional[Union[Mapping[str, Any], PathType, RandomDistortionFactoryConfig]] = None,
               out_of_path: Optional[str] = None):
        """``out_of_path``: 'raise' | 'pass_through' | None, see ``RandomDistortion`` (the reference's ``create(config)`` plus the
        one switch this path needs: what a drawn ``jpeg_quality`` does)."""
        config = dyn_structure(config, RandomDistortionFactoryConfig, support_path_type=True, support_none_type=True)

        photometric_policies, photometric_weights = self.create_policies_and_policy_weights(
            self.photometric_policy_factories, self.photometric_policy_default_weights, config)
        stage_configs = [
            RandomDistortionStageConfig(
                distortion_policies=photometric_policies,
                distortion_policy_weights=photometric_weights,
                prob_enable=config.prob_photometric,
                num_distortions_min=config.num_photometric_min,
                num_distortions_max=config.num_photometric_max,
                conflict_control_keyword_groups=config.photometric_conflict_control_keyword_groups,
            )
        ]

        geometric_policies, geometric_weights = self.create_policies_and_policy_weights(
            self.geometric_policy_factories, self.geometric_policy_default_weights, config)
        post_rotate_policy = None
        if config.force_post_rotate:
            idx = next(i for i, policy in enumerate(geometric_policies) if policy.name == 'rotate')
            post_rotate_policy = geometric_policies.pop(idx)
            geometric_weights.pop(idx)
        stage_configs.append(
            RandomDistortionStageConfig(
                distortion_policies=geometric_policies,
                distortion_policy_weights=geometric_weights,
                prob_enable=config.prob_geometric,
                num_distortions_min=1,
                num_distortions_max=1,
                inject_corner_points=config.force_post_rotate,
            ))
        if post_rotate_policy:
            stage_configs.append(
                RandomDistortionStageConfig(
                    distortion_policies=[post_rotate_policy],
                    distortion_policy_weights=[1.0],
                    prob_enable=1.0,
                    num_distortions_min=1,
                    num_distortions_max=1,
                    force_sample_level_in_full_range=True,
                ))
        return RandomDistortion(configs=stage_configs, level_min=config.level_min, level_max=config.level_max, out_of_path=out_of_path)


random_distortion_factory = RandomDistortionFactory()
