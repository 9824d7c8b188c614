"""Config generators of the affine family (reference: distortion_policy/geometric/affine.py)."""
from typing import Tuple

import attrs
from numpy.random import Generator as RandomGenerator

from vkit_amd.mechanism import distortion
from ..opt import sample_float, sample_int
from ..type import DistortionConfigGenerator, DistortionPolicyFactory


@attrs.define
class ShearHoriConfigGeneratorConfig:
    angle_min: int = 1
    angle_max: int = 30
    prob_negative: float = 0.5


class ShearHoriConfigGenerator(DistortionConfigGenerator[ShearHoriConfigGeneratorConfig, distortion.ShearHoriConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.ShearHoriConfig(angle=sample_int(self.level, self.config.angle_min, self.config.angle_max,
                                                           self.config.prob_negative, rng))


shear_hori_policy_factory = DistortionPolicyFactory(distortion.shear_hori, ShearHoriConfigGenerator)


@attrs.define
class ShearVertConfigGeneratorConfig:
    angle_min: int = 1
    angle_max: int = 30
    prob_negative: float = 0.5


class ShearVertConfigGenerator(DistortionConfigGenerator[ShearVertConfigGeneratorConfig, distortion.ShearVertConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.ShearVertConfig(angle=sample_int(self.level, self.config.angle_min, self.config.angle_max,
                                                           self.config.prob_negative, rng))


shear_vert_policy_factory = DistortionPolicyFactory(distortion.shear_vert, ShearVertConfigGenerator)


@attrs.define
class RotateConfigGeneratorConfig:
    angle_min: int = 1
    angle_max: int = 180
    prob_negative: float = 0.5


class RotateConfigGenerator(DistortionConfigGenerator[RotateConfigGeneratorConfig, distortion.RotateConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.RotateConfig(angle=sample_int(self.level, self.config.angle_min, self.config.angle_max,
                                                        self.config.prob_negative, rng))


rotate_policy_factory = DistortionPolicyFactory(distortion.rotate, RotateConfigGenerator)


def _sample_skew_ratio(generator, rng: RandomGenerator):
    ratio = sample_float(generator.level, generator.config.ratio_min, generator.config.ratio_max, None, rng)
    if rng.random() < generator.config.prob_negative:
        ratio *= -1
    return ratio


@attrs.define
class SkewHoriConfigGeneratorConfig:
    ratio_min: float = 0.0
    ratio_max: float = 0.35
    prob_negative: float = 0.5


class SkewHoriConfigGenerator(DistortionConfigGenerator[SkewHoriConfigGeneratorConfig, distortion.SkewHoriConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.SkewHoriConfig(ratio=_sample_skew_ratio(self, rng))


skew_hori_policy_factory = DistortionPolicyFactory(distortion.skew_hori, SkewHoriConfigGenerator)


@attrs.define
class SkewVertConfigGeneratorConfig:
    ratio_min: float = 0.0
    ratio_max: float = 0.35
    prob_negative: float = 0.5


class SkewVertConfigGenerator(DistortionConfigGenerator[SkewVertConfigGeneratorConfig, distortion.SkewVertConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.SkewVertConfig(ratio=_sample_skew_ratio(self, rng))


skew_vert_policy_factory = DistortionPolicyFactory(distortion.skew_vert, SkewVertConfigGenerator)
