"""Config generators of ``gaussion_noise``, ``impulse_noise`` and ``speckle_noise`` (reference:
distortion_policy/photometric/noise.py:24-52, 77-145)."""
from typing import Tuple

import attrs
from numpy.random import Generator as RandomGenerator

from vkit_amd.mechanism import distortion
from ..opt import sample_float
from ..type import DistortionConfigGenerator, DistortionPolicyFactory


@attrs.define
class GaussionNoiseConfigGeneratorConfig:
    std_min: float = 1.0
    std_max: float = 35.0


class GaussionNoiseConfigGenerator(
        DistortionConfigGenerator[GaussionNoiseConfigGeneratorConfig, distortion.GaussionNoiseConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.GaussionNoiseConfig(
            std=sample_float(self.level, self.config.std_min, self.config.std_max, None, rng))


gaussion_noise_policy_factory = DistortionPolicyFactory(distortion.gaussion_noise, GaussionNoiseConfigGenerator)


@attrs.define
class ImpulseNoiseConfigGeneratorConfig:
    prob_presv_min: float = 0.95
    prob_presv_max: float = 1.0


class ImpulseNoiseConfigGenerator(
        DistortionConfigGenerator[ImpulseNoiseConfigGeneratorConfig, distortion.ImpulseNoiseConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        prob_presv = sample_float(self.level, self.config.prob_presv_min, self.config.prob_presv_max, None, rng,
                                  inverse_level=True)
        prob_not_presv = 1 - prob_presv
        salt_ratio = rng.uniform()
        prob_salt = prob_not_presv * salt_ratio
        prob_pepper = prob_not_presv - prob_salt
        return distortion.ImpulseNoiseConfig(prob_salt=prob_salt, prob_pepper=prob_pepper)


impulse_noise_policy_factory = DistortionPolicyFactory(distortion.impulse_noise, ImpulseNoiseConfigGenerator)


@attrs.define
class SpeckleNoiseConfigGeneratorConfig:
    std_min: float = 0.0
    std_max: float = 0.3


class SpeckleNoiseConfigGenerator(
        DistortionConfigGenerator[SpeckleNoiseConfigGeneratorConfig, distortion.SpeckleNoiseConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.SpeckleNoiseConfig(
            std=sample_float(self.level, self.config.std_min, self.config.std_max, None, rng))


speckle_noise_policy_factory = DistortionPolicyFactory(distortion.speckle_noise, SpeckleNoiseConfigGenerator)


@attrs.define
class PoissonNoiseConfigGeneratorConfig:
    pass


class PoissonNoiseConfigGenerator(
        DistortionConfigGenerator[PoissonNoiseConfigGeneratorConfig, distortion.PoissonNoiseConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.PoissonNoiseConfig()


poisson_noise_policy_factory = DistortionPolicyFactory(distortion.poisson_noise, PoissonNoiseConfigGenerator)

