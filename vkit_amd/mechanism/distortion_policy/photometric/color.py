"""Config generators of ``mean_shift`` and ``color_shift`` (reference: distortion_policy/photometric/color.py:25-104)."""
from typing import Tuple

import attrs
from numpy.random import Generator as RandomGenerator

from vkit_amd.mechanism import distortion
from ..opt import sample_channels, sample_int
from ..type import DistortionConfigGenerator, DistortionPolicyFactory


@attrs.define
class MeanShiftConfigGeneratorConfig:
    delta_max: int = 127
    prob_negative: float = 0.5
    prob_enable_threshold: float = 0.5
    threshold_ratio_min: float = 1.0
    threshold_ratio_max: float = 1.5


class MeanShiftConfigGenerator(DistortionConfigGenerator[MeanShiftConfigGeneratorConfig, distortion.MeanShiftConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        cfg = self.config
        delta = sample_int(self.level, 0, cfg.delta_max, cfg.prob_negative, rng)
        channels = sample_channels(rng)
        threshold = None
        if rng.random() < cfg.prob_enable_threshold:
            ratio = rng.uniform(cfg.threshold_ratio_min, cfg.threshold_ratio_max)
            threshold = round(-delta * ratio) if delta < 0 else round(255 - delta * ratio)
        return distortion.MeanShiftConfig(delta=delta, channels=channels, threshold=threshold)


mean_shift_policy_factory = DistortionPolicyFactory(distortion.mean_shift, MeanShiftConfigGenerator)


@attrs.define
class ColorShiftConfigGeneratorConfig:
    delta_max: int = 127
    prob_negative: float = 0.5


class ColorShiftConfigGenerator(DistortionConfigGenerator[ColorShiftConfigGeneratorConfig, distortion.ColorShiftConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.ColorShiftConfig(
            delta=sample_int(self.level, 0, self.config.delta_max, self.config.prob_negative, rng))


color_shift_policy_factory = DistortionPolicyFactory(distortion.color_shift, ColorShiftConfigGenerator)
