"""Shared helpers of the photometric distortions (reference: photometric/opt.py)."""
import logging
import os
import threading
from enum import Enum, unique

from vkit_amd.element import Image, ImageMode

logger = logging.getLogger(__name__)


@unique
class OutOfBoundBehavior(Enum):
    CLIP = 'clip'
    CYCLE = 'cycle'


def to_rgb_image(image: Image, mode: ImageMode):
    if mode not in (ImageMode.GRAYSCALE, ImageMode.RGB):
        image = image.to_rgb_image()
    return image


def to_original_image(image: Image, mode: ImageMode):
    if mode not in (ImageMode.GRAYSCALE, ImageMode.RGB):
        image = image.to_target_mode_image(mode)
    return image


# Operators the reference has but whose pixel work lies outside this path (a JPEG codec round trip).
# They keep their name, config class and config generator, so that RandomDistortion's policy table, its sampling and the caller's
# rng stream stay the reference's draw for draw.  What happens to the IMAGE is an explicit choice:
#   'pass_through' (default)  the image is returned unchanged, one warning is logged per operator, and every result that went
#                             through such an operator says so: ``DistortionResult.meta['out_of_path'] = ('jpeg_quality',)``
#                             (RandomDistortion collects the names of all its stages there)
#   'raise'                   NotImplementedError: for callers that must not miss a stage
# chosen, in this order, by ``random_distortion_factory.create(config, out_of_path='raise')`` / ``RandomDistortion(...,
# out_of_path='raise')`` (the object's own setting wins inside its ``distort``), by ``with out_of_path('raise'):`` around the call,
# by the environment (VKX_OUT_OF_PATH=raise|pass_through; VKX_STRICT_UNSUPPORTED=1 is the older spelling of 'raise').
OUT_OF_PATH_OPERATORS = ('jpeg_quality',)
_warned = set()
_choice = threading.local()


def out_of_path_behaviour() -> str:
    chosen = getattr(_choice, 'value', None)
    if chosen:
        return chosen
    env = os.environ.get('VKX_OUT_OF_PATH', '')
    if env in ('raise', 'pass_through'):
        return env
    return 'raise' if os.environ.get('VKX_STRICT_UNSUPPORTED', '') == '1' else 'pass_through'


def out_of_path_context_choice():
    """What a surrounding ``with out_of_path(...)`` of this thread chose, or None."""
    return getattr(_choice, 'value', None)


class out_of_path:
    """``with out_of_path('raise'):`` / ``with out_of_path('pass_through'):`` -- the behaviour for the calls inside (this thread)."""

    def __init__(self, behaviour):
        if behaviour not in (None, 'raise', 'pass_through'):
            raise ValueError(f"out_of_path={behaviour!r}: 'raise' or 'pass_through'")
        self.behaviour = behaviour

    def __enter__(self):
        self.prev = getattr(_choice, 'value', None)
        if self.behaviour is not None:
            _choice.value = self.behaviour
        return self

    def __exit__(self, *exc):
        _choice.value = self.prev
        return False


def take_passed_through():
    """Names of the out-of-path operators that passed an image through since the last call (this thread); clears the list."""
    names = getattr(_choice, 'passed', None) or []
    _choice.passed = []
    return tuple(names)


def pass_through_out_of_path(name: str, image: Image) -> Image:
    if out_of_path_behaviour() == 'raise':
        raise NotImplementedError(
            f'distortion "{name}" is not part of the MI355X-accelerated path and out_of_path is \'raise\' '
            f'(photometric/opt.py: out_of_path)')
    if name not in _warned:
        _warned.add(name)
        logger.warning('distortion "%s" is outside the accelerated path: its config is sampled like the reference\'s, '
                       'the image passes through unchanged (DistortionResult.meta[\'out_of_path\'] records it)', name)
    passed = getattr(_choice, 'passed', None)
    if passed is None:
        passed = _choice.passed = []
    passed.append(name)
    return image
