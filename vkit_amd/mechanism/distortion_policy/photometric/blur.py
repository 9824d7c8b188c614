"""Config generator of ``gaussian_blur`` (reference: distortion_policy/photometric/blur.py:25-53)."""
from typing import Tuple

import attrs
from numpy.random import Generator as RandomGenerator

from vkit_amd.mechanism import distortion
from ..opt import sample_float
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
