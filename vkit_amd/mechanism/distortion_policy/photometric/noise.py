"""Config generator of ``gaussion_noise`` (reference: distortion_policy/photometric/noise.py:24-52)."""
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
