"""Config generator of ``similarity_mls`` (reference: distortion_policy/geometric/mls.py): a lattice of handle
points whose spacings are shuffled, each handle displaced by a level-dependent integer radius."""
from typing import List, Tuple

import attrs
from numpy.random import Generator as RandomGenerator

from vkit_amd.element import Point, PointList
from vkit_amd.mechanism import distortion
from ..opt import SampleFloatMode, generate_grid_size, sample_float
from ..type import DistortionConfigGenerator, DistortionPolicyFactory


@attrs.define
class SimilarityMlsConfigGeneratorConfig:
    num_segments_min: int = 2
    num_segments_max: int = 4
    step_min: int = 10
    radius_max_ratio_min: float = 0.025
    radius_max_ratio_max: float = 0.125
    grid_size_min: int = 15
    grid_size_ratio: float = 0.01


class SimilarityMlsConfigGenerator(
        DistortionConfigGenerator[SimilarityMlsConfigGeneratorConfig, distortion.SimilarityMlsConfig]):

    @classmethod
    def generate_coord(cls, length: int, step: int, rng: RandomGenerator):
        """0 .. length-1 split into shuffled segments of ``step`` (the remainder joins the last one)."""
        end = length - 1
        if end % step == 0:
            steps = [step] * (end // step)
        else:
            steps = [step] * (end // step - 1)
            steps.append(step + end % step)
        assert sum(steps) == end
        rng.shuffle(steps)
        coord: List[int] = [0]
        for seg in steps:
            coord.append(coord[-1] + seg)
        return coord

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        height, width = shape
        short_side = min(shape)
        num_segments = rng.integers(self.config.num_segments_min, self.config.num_segments_max + 1)
        step = (short_side - 1) // num_segments
        if step < self.config.step_min:
            step = short_side - 1  # too dense: corners only

        coord_y = self.generate_coord(height, step, rng)
        coord_x = self.generate_coord(width, step, rng)
        src_handle_points = PointList(Point.create(y=y, x=x) for y in coord_y for x in coord_x)

        assert self.config.radius_max_ratio_max < 0.5
        radius_max_ratio = sample_float(self.level, self.config.radius_max_ratio_min,
                                        self.config.radius_max_ratio_max, None, rng, mode=SampleFloatMode.QUAD)
        radius = int(radius_max_ratio * step)
        dst_handle_points = PointList()
        for point in src_handle_points:
            delta_y = rng.integers(-radius, radius + 1)
            delta_x = rng.integers(-radius, radius + 1)
            dst_handle_points.append(Point.create(y=point.y + delta_y, x=point.x + delta_x))

        return distortion.SimilarityMlsConfig(
            src_handle_points=src_handle_points.to_point_tuple(),
            dst_handle_points=dst_handle_points.to_point_tuple(),
            grid_size=generate_grid_size(self.config.grid_size_min, self.config.grid_size_ratio, shape),
        )


similarity_mls_policy_factory = DistortionPolicyFactory(distortion.similarity_mls, SimilarityMlsConfigGenerator)
