"""Config generator of ``similarity_mls`` (reference: distortion_policy/geometric/mls.py): a lattice of handle
points whose spacings are shuffled, each handle displaced by a level-dependent integer radius."""
import itertools
from typing import Tuple

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
        """Cut positions of ``0 .. length - 1``: whole ``step`` segments (the remainder is added to one of them), in a
        shuffled order.  One ``rng.shuffle`` of a Python list, like the reference (:48-64)."""
        end = length - 1
        whole, rest = divmod(end, step)
        segments = [step] * whole
        if rest:
            segments[-1] += rest
        assert sum(segments) == end
        rng.shuffle(segments)
        return list(itertools.accumulate(segments, initial=0))

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        cfg = self.config
        # draw order (the contract): segment count, row shuffle, column shuffle, radius ratio, then (dy, dx) per handle
        num_segments = rng.integers(cfg.num_segments_min, cfg.num_segments_max + 1)
        step = (min(shape) - 1) // num_segments
        if step < cfg.step_min:
            step = min(shape) - 1          # too dense for this page: only the corners stay
        rows = self.generate_coord(shape[0], step, rng)
        cols = self.generate_coord(shape[1], step, rng)
        handles = PointList(Point.create(y=y, x=x) for y, x in itertools.product(rows, cols))

        assert cfg.radius_max_ratio_max < 0.5       # neighbouring handles cannot cross
        ratio = sample_float(self.level, cfg.radius_max_ratio_min, cfg.radius_max_ratio_max, None, rng,
                             mode=SampleFloatMode.QUAD)
        radius = int(ratio * step)

        def jitter(value: int) -> int:
            return value + rng.integers(-radius, radius + 1)

        moved = PointList(Point.create(y=jitter(p.y), x=jitter(p.x)) for p in handles)      # y is drawn before x
        return distortion.SimilarityMlsConfig(
            src_handle_points=handles.to_point_tuple(), dst_handle_points=moved.to_point_tuple(),
            grid_size=generate_grid_size(cfg.grid_size_min, cfg.grid_size_ratio, shape))


similarity_mls_policy_factory = DistortionPolicyFactory(distortion.similarity_mls, SimilarityMlsConfigGenerator)
