"""Config generators of ``line_streak`` and ``rectangle_streak`` (reference:
distortion_policy/photometric/streak.py:25-247)."""
from typing import Tuple

import attrs
from numpy.random import Generator as RandomGenerator

from vkit_amd.mechanism import distortion
from ..opt import sample_float
from ..type import DistortionConfigGenerator, DistortionPolicyFactory


def _sample_dash(cfg, long_side_length: int, rng: RandomGenerator):
    """(dash_thickness, dash_gap); one rng.random() gate, then two uniforms when dashed."""
    if not rng.random() < cfg.prob_dash:
        return 0, 0
    thickness_ratio = float(rng.uniform(cfg.dash_thickness_ratio_min, cfg.dash_thickness_ratio_max))
    dash_thickness = round(thickness_ratio * long_side_length)
    gap_ratio = float(rng.uniform(cfg.dash_to_thickness_gap_ratio_min, cfg.dash_to_thickness_gap_ratio_max))
    return dash_thickness, round(gap_ratio * dash_thickness)


@attrs.define
class LineStreakConfigGeneratorConfig:
    thickness_min: int = 1
    thickness_max: int = 4
    gap_min: int = 5
    gap_ratio_min: float = 0.01
    gap_ratio_max: float = 0.5
    prob_dash: float = 0.25
    dash_thickness_ratio_min: float = 0.0
    dash_thickness_ratio_max: float = 0.05
    dash_to_thickness_gap_ratio_min: float = 0.5
    dash_to_thickness_gap_ratio_max: float = 1.0
    alpha_min: float = 0.2
    alpha_max: float = 1.0


class LineStreakConfigGenerator(DistortionConfigGenerator[LineStreakConfigGeneratorConfig, distortion.LineStreakConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        cfg = self.config
        long_side_length = max(shape)
        gap_ratio = sample_float(self.level, cfg.gap_ratio_min, cfg.gap_ratio_max, None, rng, inverse_level=True)
        gap = max(cfg.gap_min, round(gap_ratio * long_side_length))
        thickness = rng.integers(cfg.thickness_min, cfg.thickness_max + 1)
        dash_thickness, dash_gap = _sample_dash(cfg, long_side_length, rng)
        alpha = rng.uniform(cfg.alpha_min, cfg.alpha_max)
        mode = rng.integers(0, 3)  # 0: vertical, 1: horizontal, 2: both
        if mode not in (0, 1, 2):
            raise NotImplementedError()
        return distortion.LineStreakConfig(
            thickness=thickness,
            gap=gap,
            dash_thickness=dash_thickness,
            dash_gap=dash_gap,
            alpha=alpha,
            enable_vert=mode in (0, 2),
            enable_hori=mode in (1, 2),
        )


line_streak_policy_factory = DistortionPolicyFactory(distortion.line_streak, LineStreakConfigGenerator)


def sample_params_for_rectangle_and_ellipse_streak(level: int, thickness_min: int, thickness_max: int,
                                                   aspect_ratio_min: float, aspect_ratio_max: float,
                                                   short_side_min: int, short_side_min_ratio_min: float,
                                                   short_side_min_ratio_max: float, short_side_step_ratio_min: float,
                                                   short_side_step_ratio_max: float, alpha_min: float,
                                                   alpha_max: float, shape: Tuple[int, int], rng: RandomGenerator):
    long_side_length = max(shape)
    ratio = sample_float(level, short_side_min_ratio_min, short_side_min_ratio_max, None, rng, inverse_level=True)
    short_side_min = max(short_side_min, round(ratio * long_side_length))
    step_ratio = rng.uniform(short_side_step_ratio_min, short_side_step_ratio_max)
    short_side_step = round(step_ratio * short_side_min)
    thickness = rng.integers(thickness_min, thickness_max + 1)
    aspect_ratio = rng.uniform(aspect_ratio_min, aspect_ratio_max)
    alpha = rng.uniform(alpha_min, alpha_max)
    return thickness, aspect_ratio, short_side_min, short_side_step, alpha


@attrs.define
class RectangleStreakConfigGeneratorConfig:
    thickness_min: int = 1
    thickness_max: int = 4
    aspect_ratio_min: float = 0.5
    aspect_ratio_max: float = 1.5
    prob_dash: float = 0.25
    dash_thickness_ratio_min: float = 0.0
    dash_thickness_ratio_max: float = 0.05
    dash_to_thickness_gap_ratio_min: float = 0.5
    dash_to_thickness_gap_ratio_max: float = 1.0
    short_side_min: int = 5
    short_side_min_ratio_min: float = 0.01
    short_side_min_ratio_max: float = 0.25
    short_side_step_ratio_min: float = 0.8
    short_side_step_ratio_max: float = 3.0
    alpha_min: float = 0.2
    alpha_max: float = 1.0


class RectangleStreakConfigGenerator(
        DistortionConfigGenerator[RectangleStreakConfigGeneratorConfig, distortion.RectangleStreakConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        cfg = self.config
        thickness, aspect_ratio, short_side_min, short_side_step, alpha = \
            sample_params_for_rectangle_and_ellipse_streak(
                self.level, cfg.thickness_min, cfg.thickness_max, cfg.aspect_ratio_min, cfg.aspect_ratio_max,
                cfg.short_side_min, cfg.short_side_min_ratio_min, cfg.short_side_min_ratio_max,
                cfg.short_side_step_ratio_min, cfg.short_side_step_ratio_max, cfg.alpha_min, cfg.alpha_max, shape, rng)
        dash_thickness, dash_gap = _sample_dash(cfg, max(shape), rng)
        return distortion.RectangleStreakConfig(
            thickness=thickness,
            aspect_ratio=aspect_ratio,
            dash_thickness=dash_thickness,
            dash_gap=dash_gap,
            short_side_min=short_side_min,
            short_side_step=short_side_step,
            alpha=alpha,
        )


rectangle_streak_policy_factory = DistortionPolicyFactory(distortion.rectangle_streak, RectangleStreakConfigGenerator)


@attrs.define
class EllipseStreakConfigGeneratorConfig:
    thickness_min: int = 1
    thickness_max: int = 3
    aspect_ratio_min: float = 0.5
    aspect_ratio_max: float = 1.5
    short_side_min: int = 5
    short_side_min_ratio_min: float = 0.01
    short_side_min_ratio_max: float = 0.25
    short_side_step_ratio_min: float = 0.8
    short_side_step_ratio_max: float = 3.0
    alpha_min: float = 0.2
    alpha_max: float = 1.0


class EllipseStreakConfigGenerator(
        DistortionConfigGenerator[EllipseStreakConfigGeneratorConfig, distortion.EllipseStreakConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        cfg = self.config
        thickness, aspect_ratio, short_side_min, short_side_step, alpha = \
            sample_params_for_rectangle_and_ellipse_streak(
                self.level, cfg.thickness_min, cfg.thickness_max, cfg.aspect_ratio_min, cfg.aspect_ratio_max,
                cfg.short_side_min, cfg.short_side_min_ratio_min, cfg.short_side_min_ratio_max,
                cfg.short_side_step_ratio_min, cfg.short_side_step_ratio_max, cfg.alpha_min, cfg.alpha_max, shape, rng)
        return distortion.EllipseStreakConfig(thickness=thickness, aspect_ratio=aspect_ratio,
                                              short_side_min=short_side_min, short_side_step=short_side_step, alpha=alpha)


ellipse_streak_policy_factory = DistortionPolicyFactory(distortion.ellipse_streak, EllipseStreakConfigGenerator)
