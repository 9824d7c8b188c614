"""PageResizingStep: the step right after page distortion (reference: vkit/pipeline/text_detection/page_resizing.py:110-181).

The page is rescaled so that its smallest (outlier-filtered) text-line height lands on a sampled target height.  One
sampled cv2 interpolation (``sample_cv_resize_interpolation``: NEAREST_EXACT, LINEAR_EXACT, CUBIC, LANCZOS4, plus AREA
when shrinking) serves the image, the four masks and the two height score maps; score values scale with the page.
Every resize is a device kernel of ``csrc/resize.hip`` behind ``to_resized_image / _mask / _score_map``.
"""
from typing import Sequence

import attrs
import numpy as np
from numpy.random import Generator as RandomGenerator

from vkit_amd import _native
from vkit_amd.element import Image, Mask, ScoreMap
from vkit_amd.utility import sample_cv_resize_interpolation
from ..interface import PipelineStep, PipelineStepFactory
from .page_distortion import PageDistortionStepOutput


@attrs.define
class PageResizingStepConfig:
    resized_text_line_height_min: float = 3.0
    resized_text_line_height_max: float = 10.0
    text_line_heights_filtering_thr: float = 1.0


@attrs.define
class PageResizingStepInput:
    page_distortion_step_output: PageDistortionStepOutput


@attrs.define
class PageResizingStepOutput:
    page_image: Image
    page_active_mask: Mask
    page_char_mask: Mask
    page_seal_impression_char_mask: Mask
    page_char_height_score_map: ScoreMap
    page_text_line_mask: Mask
    page_text_line_height_score_map: ScoreMap


# the elements in the order the reference resizes them (page_resizing.py:123-170)
_ELEMENTS = (('page_active_mask', 'mask'), ('page_char_mask', 'mask'), ('page_seal_impression_char_mask', 'mask'),
             ('page_char_height_score_map', 'score_map'), ('page_text_line_mask', 'mask'),
             ('page_text_line_height_score_map', 'score_map'))


class PageResizingStep(PipelineStep[PageResizingStepConfig, PageResizingStepInput, PageResizingStepOutput]):

    def get_text_line_heights_min(self, page_distorted_text_line_heights: Sequence[float]):
        """Smallest height above the filtering threshold whose distance from the median is below 3.5 median absolute
        deviations (reference page_resizing.py:63-84)."""
        heights = np.asarray([h for h in page_distorted_text_line_heights
                              if h > self.config.text_line_heights_filtering_thr])
        assert heights.size
        deviation = np.abs(heights - np.median(heights))
        ratio = deviation / (np.median(deviation) or 1.0)
        return float(min(h for h, r in zip(heights, ratio) if r < 3.5))

    def run(self, input: PageResizingStepInput, rng: RandomGenerator):
        src = input.page_distortion_step_output
        for name, _ in _ELEMENTS:
            assert getattr(src, name) is not None, name
        assert src.page_text_line_heights

        height, width = src.page_image.shape
        # rng order is part of the contract: target height first, then the interpolation
        target = rng.uniform(self.config.resized_text_line_height_min, self.config.resized_text_line_height_max)
        resize_ratio = target / self.get_text_line_heights_min(src.page_text_line_heights)
        resized_height, resized_width = round(resize_ratio * height), round(resize_ratio * width)
        interpolation = sample_cv_resize_interpolation(rng, include_cv_inter_area=(resize_ratio < 1.0))
        size = dict(resized_height=resized_height, resized_width=resized_width, cv_resize_interpolation=interpolation)

        out = {'page_image': src.page_image.to_resized_image(**size)}
        # the four masks, where they live on the device: Mask.to_resized_mask's three steps -- (> 0) * 255, cv.resize, > threshold -- with
        # the two look-ups of ALL masks in one launch each (eight launches of a few microseconds as two; same planes)
        masks = [(name, getattr(src, name)) for name, kind in _ELEMENTS if kind == 'mask']
        if all(getattr(m, 'on_device', False) and not m.box and m.shape == (height, width) for _n, m in masks):
            from vkit_amd.element.mask import _LUT_X255
            from vkit_amd.element.opt import generate_resized_shape
            mask_shape = generate_resized_shape(height, width, resized_height, resized_width)      # (raises like to_resized_mask)
            spread = _native.apply_lut_planes([m.arr for _n, m in masks], [_LUT_X255] * len(masks))
            resized_planes = [_native.resize(plane, mask_shape, interpolation) for plane in spread]
            above = (np.arange(256) > 0).astype(np.uint8)
            for (name, _m), plane in zip(masks, _native.apply_lut_planes(resized_planes, [above] * len(masks))):
                out[name] = Mask(mat=plane)
        for name, kind in _ELEMENTS:
            element = getattr(src, name)
            assert element.shape == (height, width), name
            if name in out:
                continue
            if kind == 'mask':
                out[name] = element.to_resized_mask(**size)
            else:
                resized = element.to_resized_score_map(**size)
                resized.assign_mat(resized.mat * resize_ratio)      # heights shrink / grow with the page
                out[name] = resized
        return PageResizingStepOutput(**out)


page_resizing_step_factory = PipelineStepFactory(PageResizingStep)
