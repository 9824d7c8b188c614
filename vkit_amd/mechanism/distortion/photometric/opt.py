"""Shared helpers of the photometric distortions (reference: photometric/opt.py)."""
import logging
import os
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
# They keep their name, config class and config generator, so that RandomDistortion's
# policy table, its sampling and the caller's rng stream stay the reference's draw for draw; the image passes through
# unchanged and a warning is logged once per operator.  VKX_STRICT_UNSUPPORTED=1 turns the pass-through into a
# NotImplementedError for callers that must not miss a stage silently.
OUT_OF_PATH_OPERATORS = ('jpeg_quality',)
_warned = set()


def pass_through_out_of_path(name: str, image: Image) -> Image:
    if os.environ.get('VKX_STRICT_UNSUPPORTED', '') == '1':
        raise NotImplementedError(
            f'distortion "{name}" is not part of the MI355X-accelerated path (VKX_STRICT_UNSUPPORTED=1)')
    if name not in _warned:
        _warned.add(name)
        logger.warning('distortion "%s" is outside the accelerated path: its config is sampled like the reference\'s, '
                       'the image passes through unchanged', name)
    return image
