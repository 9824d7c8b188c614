"""Config generators of ``gaussian_blur``, ``glass_blur`` and ``zoom_in_blur`` (reference:
distortion_policy/photometric/blur.py:25-53, 121-219)."""
from typing import Tuple

import attrs
from numpy.random import Generator as RandomGenerator

from vkit_amd.mechanism import distortion
from ..opt import sample_float, sample_int
from ..type import DistortionConfigGenerator, DistortionPolicyFactory


@attrs.define
class GaussianBlurConfigGeneratorConfig:
    sigma_min: float = 0.5
    sigma_max: float = 1.0


class GaussianBlurConfigGenerator(
        DistortionConfigGenerator[GaussianBlurConfigGeneratorConfig, distortion.GaussianBlurConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.GaussianBlurConfig(
            sigma=sample_float(self.level, self.config.sigma_min, self.config.sigma_max, None, rng))


gaussian_blur_policy_factory = DistortionPolicyFactory(distortion.gaussian_blur, GaussianBlurConfigGenerator)


@attrs.define
class DefocusBlurConfigGeneratorConfig:
    radius_min: int = 1
    radius_max: int = 2


class DefocusBlurConfigGenerator(
        DistortionConfigGenerator[DefocusBlurConfigGeneratorConfig, distortion.DefocusBlurConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.DefocusBlurConfig(
            radius=sample_int(self.level, self.config.radius_min, self.config.radius_max, None, rng))


defocus_blur_policy_factory = DistortionPolicyFactory(distortion.defocus_blur, DefocusBlurConfigGenerator)


@attrs.define
class MotionBlurConfigGeneratorConfig:
    radius_min: int = 1
    radius_max: int = 2


class MotionBlurConfigGenerator(
        DistortionConfigGenerator[MotionBlurConfigGeneratorConfig, distortion.MotionBlurConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        radius = sample_int(self.level, self.config.radius_min, self.config.radius_max, None, rng)
        return distortion.MotionBlurConfig(radius=radius, angle=rng.integers(0, 360))


motion_blur_policy_factory = DistortionPolicyFactory(distortion.motion_blur, MotionBlurConfigGenerator)


@attrs.define
class GlassBlurConfigGeneratorConfig:
    sigma_min: float = 0.5
    sigma_max: float = 1.0
    delta_min: int = 1
    delta_max: int = 1
    loop_min: int = 1
    loop_max: int = 4


class GlassBlurConfigGenerator(DistortionConfigGenerator[GlassBlurConfigGeneratorConfig, distortion.GlassBlurConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        cfg = self.config
        sigma = sample_float(self.level, cfg.sigma_min, cfg.sigma_max, None, rng)
        delta = sample_int(self.level, cfg.delta_min, cfg.delta_max, None, rng)
        loop = sample_int(self.level, cfg.loop_min, cfg.loop_max, None, rng)
        return distortion.GlassBlurConfig(sigma=sigma, delta=delta, loop=loop)


glass_blur_policy_factory = DistortionPolicyFactory(distortion.glass_blur, GlassBlurConfigGenerator)


@attrs.define
class ZoomInBlurConfigGeneratorConfig:
    ratio_min: float = 0.01
    ratio_max: float = 0.1
    step_min: float = 0.002
    step_max: float = 0.02
    alpha_min: float = 0.5
    alpha_max: float = 0.7


class ZoomInBlurConfigGenerator(DistortionConfigGenerator[ZoomInBlurConfigGeneratorConfig, distortion.ZoomInBlurConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        cfg = self.config
        ratio = sample_float(self.level, cfg.ratio_min, cfg.ratio_max, None, rng)
        step = sample_float(self.level, cfg.step_min, cfg.step_max, None, rng)
        alpha = rng.uniform(cfg.alpha_min, cfg.alpha_max)
        return distortion.ZoomInBlurConfig(ratio=ratio, step=step, alpha=alpha)


zoom_in_blur_policy_factory = DistortionPolicyFactory(distortion.zoom_in_blur, ZoomInBlurConfigGenerator)

