"""Config generators of ``jpeg_quality``, ``pixelation`` and ``fog`` (reference:
distortion_policy/photometric/effect.py:24-130)."""
from typing import Tuple

import attrs
from numpy.random import Generator as RandomGenerator

from vkit_amd.mechanism import distortion
from ..opt import sample_float, sample_int
from ..type import DistortionConfigGenerator, DistortionPolicyFactory


@attrs.define
class JpegQualityConfigGeneratorConfig:
    quality_min: int = 1
    quality_max: int = 50


class JpegQualityConfigGenerator(DistortionConfigGenerator[JpegQualityConfigGeneratorConfig, distortion.JpegQualityConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.JpegQualityConfig(
            quality=sample_int(self.level, self.config.quality_min, self.config.quality_max, None, rng, inverse_level=True))


jpeg_quality_policy_factory = DistortionPolicyFactory(distortion.jpeg_quality, JpegQualityConfigGenerator)


@attrs.define
class FogConfigGeneratorConfig:
    roughness_min: float = 0.2
    roughness_max: float = 0.85
    ratio_max_min: float = 0.2
    ratio_max_max: float = 0.75


class FogConfigGenerator(DistortionConfigGenerator[FogConfigGeneratorConfig, distortion.FogConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        roughness = sample_float(self.level, self.config.roughness_min, self.config.roughness_max, None, rng)
        ratio_max = sample_float(self.level, self.config.ratio_max_min, self.config.ratio_max_max, None, rng)
        return distortion.FogConfig(roughness=roughness, ratio_max=ratio_max)


fog_policy_factory = DistortionPolicyFactory(distortion.fog, FogConfigGenerator)


@attrs.define
class PixelationConfigGeneratorConfig:
    ratio_min: float = 0.3
    ratio_max: float = 1.0


class PixelationConfigGenerator(DistortionConfigGenerator[PixelationConfigGeneratorConfig, distortion.PixelationConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.PixelationConfig(
            ratio=sample_float(self.level, self.config.ratio_min, self.config.ratio_max, None, rng, inverse_level=True))


pixelation_policy_factory = DistortionPolicyFactory(distortion.pixelation, PixelationConfigGenerator)

