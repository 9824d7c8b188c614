"""Shared helpers of the photometric distortions (reference: photometric/opt.py)."""
from enum import Enum, unique

from vkit_amd.element import Image, ImageMode


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
