"""Level -> parameter samplers shared by the config generators (reference: distortion_policy/opt.py).
The ORDER and KIND of rng draws is part of the contract: a given Generator state must yield the reference's
configs."""
from enum import Enum, auto
from typing import Optional, Tuple

from numpy.random import Generator as RandomGenerator

from vkit_amd.utility import rng_choice_with_size

LEVEL_MIN = 1
LEVEL_MAX = 10
CHANNELS = [0, 1, 2]


def sample_channels(rng: RandomGenerator):
    num_channels = rng.integers(1, 4)
    if num_channels < 3:
        return sorted(rng_choice_with_size(rng, CHANNELS, num_channels, replace=False))
    return None


def sample_int(level: int, value_min: int, value_max: int, prob_negative: Optional[float], rng: RandomGenerator,
               inverse_level: bool = False):
    """Uniform integer from the level's slice of [value_min, value_max]; optionally negated."""
    if inverse_level:
        level = LEVEL_MAX + 1 - level
    span = value_max - value_min
    low = round(value_min + (level - 1) / LEVEL_MAX * span)
    high = round(value_min + level / LEVEL_MAX * span)
    if level == LEVEL_MAX:
        high += 1  # value_max itself must be reachable
    value = rng.integers(low, max(low + 1, high))
    if prob_negative and rng.random() < prob_negative:
        value *= -1
    return int(value)


class SampleFloatMode(Enum):
    LINEAR = auto()
    QUAD = auto()


def func_quad(x: float):
    return -x**2 + 2 * x


def sample_float(level: int, value_min: float, value_max: float, prob_reciprocal: Optional[float],
                 rng: RandomGenerator, mode: SampleFloatMode = SampleFloatMode.LINEAR, inverse_level: bool = False):
    """Uniform float from the level's slice of [value_min, value_max] (linear or ease-out quadratic slicing)."""
    if inverse_level:
        level = LEVEL_MAX + 1 - level
    span = value_max - value_min
    if mode == SampleFloatMode.LINEAR:
        ratio_low, ratio_high = (level - 1) / LEVEL_MAX, level / LEVEL_MAX
    elif mode == SampleFloatMode.QUAD:
        ratio_low, ratio_high = func_quad((level - 1) / LEVEL_MAX), func_quad(level / LEVEL_MAX)
    else:
        raise NotImplementedError()
    value = rng.uniform(value_min + ratio_low * span, value_min + ratio_high * span)
    if prob_reciprocal and rng.random() < prob_reciprocal:
        value = 1 / value
    return value


def generate_grid_size(grid_size_min: int, grid_size_ratio: float, shape: Tuple[int, int]):
    return max(grid_size_min, int(grid_size_ratio * max(shape)))
