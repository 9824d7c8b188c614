"""Config generators of ``mean_shift``, ``color_shift``, ``complement``, ``posterization`` and ``channel_permutation``
(reference: distortion_policy/photometric/color.py:25-104, 221-284, 319-338)."""
from typing import Tuple

import attrs
from numpy.random import Generator as RandomGenerator

from vkit_amd.mechanism import distortion
from ..opt import LEVEL_MAX, sample_channels, sample_float, sample_int
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


@attrs.define
class ComplementConfigGeneratorConfig:
    enable_threshold_level: int = 6
    threshold_min: int = 77
    threshold_max: int = 177


class ComplementConfigGenerator(DistortionConfigGenerator[ComplementConfigGeneratorConfig, distortion.ComplementConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        channels = sample_channels(rng)
        threshold = None
        enable_threshold_lte = (rng.random() < 0.5)
        if self.level >= self.config.enable_threshold_level:
            threshold = rng.integers(self.config.threshold_min, self.config.threshold_max + 1)
        return distortion.ComplementConfig(threshold=threshold, enable_threshold_lte=enable_threshold_lte,
                                           channels=channels)


complement_policy_factory = DistortionPolicyFactory(distortion.complement, ComplementConfigGenerator)


@attrs.define
class PosterizationConfigGeneratorConfig:
    enable_threshold_level: int = 6
    threshold_min: int = 77
    threshold_max: int = 177


class PosterizationConfigGenerator(
        DistortionConfigGenerator[PosterizationConfigGeneratorConfig, distortion.PosterizationConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        num_bits = round(self.level / LEVEL_MAX * 7)   # to [1, 7]
        channels = sample_channels(rng)
        return distortion.PosterizationConfig(num_bits=num_bits, channels=channels)


posterization_policy_factory = DistortionPolicyFactory(distortion.posterization, PosterizationConfigGenerator)


@attrs.define
class ChannelPermutationConfigGeneratorConfig:
    pass


class ChannelPermutationConfigGenerator(
        DistortionConfigGenerator[ChannelPermutationConfigGeneratorConfig, distortion.ChannelPermutationConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.ChannelPermutationConfig()


channel_permutation_policy_factory = DistortionPolicyFactory(distortion.channel_permutation,
                                                             ChannelPermutationConfigGenerator)


@attrs.define
class BrightnessShiftConfigGeneratorConfig:
    delta_max: int = 127
    prob_negative: float = 0.5


class BrightnessShiftConfigGenerator(
        DistortionConfigGenerator[BrightnessShiftConfigGeneratorConfig, distortion.BrightnessShiftConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.BrightnessShiftConfig(
            delta=sample_int(self.level, 0, self.config.delta_max, self.config.prob_negative, rng))


brightness_shift_policy_factory = DistortionPolicyFactory(distortion.brightness_shift, BrightnessShiftConfigGenerator)


@attrs.define
class StdShiftConfigGeneratorConfig:
    scale_min: float = 1.0
    scale_max: float = 2.5
    prob_reciprocal: float = 0.5


class StdShiftConfigGenerator(DistortionConfigGenerator[StdShiftConfigGeneratorConfig, distortion.StdShiftConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        scale = sample_float(self.level, self.config.scale_min, self.config.scale_max, self.config.prob_reciprocal, rng)
        return distortion.StdShiftConfig(scale=scale, channels=sample_channels(rng))


std_shift_policy_factory = DistortionPolicyFactory(distortion.std_shift, StdShiftConfigGenerator)


@attrs.define
class ColorBalanceConfigGeneratorConfig:
    ratio_min: float = 0.0
    ratio_max: float = 1.0


class ColorBalanceConfigGenerator(
        DistortionConfigGenerator[ColorBalanceConfigGeneratorConfig, distortion.ColorBalanceConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.ColorBalanceConfig(
            ratio=sample_float(self.level, self.config.ratio_min, self.config.ratio_max, None, rng, inverse_level=True))


color_balance_policy_factory = DistortionPolicyFactory(distortion.color_balance, ColorBalanceConfigGenerator)


@attrs.define
class BoundaryEqualizationConfigGeneratorConfig:
    pass


class BoundaryEqualizationConfigGenerator(
        DistortionConfigGenerator[BoundaryEqualizationConfigGeneratorConfig, distortion.BoundaryEqualizationConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.BoundaryEqualizationConfig(channels=sample_channels(rng))


boundary_equalization_policy_factory = DistortionPolicyFactory(distortion.boundary_equalization,
                                                               BoundaryEqualizationConfigGenerator)


@attrs.define
class HistogramEqualizationConfigGeneratorConfig:
    pass


class HistogramEqualizationConfigGenerator(
        DistortionConfigGenerator[HistogramEqualizationConfigGeneratorConfig, distortion.HistogramEqualizationConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        return distortion.HistogramEqualizationConfig(channels=sample_channels(rng))


histogram_equalization_policy_factory = DistortionPolicyFactory(distortion.histogram_equalization,
                                                                HistogramEqualizationConfigGenerator)

