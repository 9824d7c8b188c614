"""Page assembler: the ordered layer list of a synthetic page composited onto its background
(reference: vkit/pipeline/text_detection/page_assembler.py:120-274).

Layer order (the contract, reference :155-236): background copy -> page images (box, image, scalar alpha) ->
QR barcodes, code-39 barcodes (score-map alpha, black) -> text-line bounding boxes (score-map alpha, colour) ->
text lines (score-map alpha + glyph colour, or mask + rendered image) -> non-text symbols (box, image, alpha
array or scalar) -> seal impressions (rotate, then background mask with scalar alpha, then text score map).

MI355X shape of the step: the loop only RECORDS layers (``deferred_fill``); the page crosses PCIe once and
all layers are applied on the device in that order by one ``vkx_fill_u8`` call.  The seal impressions' rotation
runs through the ``rotate`` operator (``vkx_warp_affine_*``) like in the reference.

The inputs are the outputs of upstream steps that are outside the accelerated path (layout, fonts, barcodes
...); they are declared here with the reference's field names, reduced to the fields this step reads.  One
deviation: ``fill_text_line_to_seal_impression`` (vkit/engine/seal_impression) is upstream rendering, so a
``SealImpressionResource`` carries its result (``text_line_filled_score_map``, ``char_polygons``) directly.
"""
from typing import List, Optional, Sequence, Tuple, Union

import attrs
import numpy as np
from numpy.random import Generator as RandomGenerator

from vkit_amd import _native
from vkit_amd.element import Box, Image, Mask, PointList, Polygon, ScoreMap, Shapable
from vkit_amd.element.opt import deferred_fill
from vkit_amd.mechanism.distortion import rotate
from ..interface import PipelineStep, PipelineStepFactory


# ---- upstream outputs, reduced to what the assembler reads ---------------------------------------------------
@attrs.define
class DisconnectedTextRegion:
    polygon: Polygon


@attrs.define
class NonTextRegion:
    polygon: Polygon


@attrs.define
class PageLayout:
    height: int
    width: int
    disconnected_text_regions: Sequence[DisconnectedTextRegion] = ()
    non_text_regions: Sequence[NonTextRegion] = ()


@attrs.define
class PageLayoutStepOutput:
    page_layout: PageLayout


@attrs.define
class PageBackgroundStepOutput:
    background_image: Image


@attrs.define
class PageImage:
    image: Image
    box: Box
    alpha: float


@attrs.define
class PageImageCollection:
    height: int
    width: int
    page_images: Sequence[PageImage] = ()


@attrs.define
class PageImageStepOutput:
    page_image_collection: PageImageCollection
    # For filling the inactive region caused by distortion.
    page_bottom_layer_image: Image


@attrs.define
class PageBarcodeStepOutput:
    height: int
    width: int
    barcode_qr_score_maps: Sequence[ScoreMap] = ()
    barcode_code39_score_maps: Sequence[ScoreMap] = ()


@attrs.define
class TextLine:
    """A rendered text line bound to the page (reference: vkit/engine/font/type.py:455-481): box-attached mask,
    optional box-attached score map (anti-aliased glyph coverage) and the rendered image."""
    image: Image
    mask: Mask
    score_map: Optional[ScoreMap]
    glyph_color: Tuple[int, int, int]

    @property
    def box(self):
        assert self.mask.box
        return self.mask.box


@attrs.define
class PageTextLineCollection:
    height: int
    width: int
    text_lines: Sequence[TextLine] = ()
    short_text_line_flags: Sequence[bool] = ()

    @property
    def shape(self):
        return self.height, self.width


@attrs.define
class SealImpression:
    alpha: float
    color: Tuple[int, int, int]
    background_mask: Mask


@attrs.define
class SealImpressionResource:
    box: Box
    angle: int
    text_line_filled_score_map: ScoreMap
    char_polygons: Sequence[Polygon] = ()


@attrs.define
class PageSealImpressionTextLineCollection:
    height: int
    width: int
    seal_impressions: Sequence[SealImpression] = ()
    seal_impression_resources: Sequence[SealImpressionResource] = ()


@attrs.define
class PageTextLineStepOutput:
    page_text_line_collection: PageTextLineCollection
    page_seal_impression_text_line_collection: PageSealImpressionTextLineCollection


@attrs.define
class PageNonTextSymbolStepOutput:
    images: Sequence[Image] = ()
    boxes: Sequence[Box] = ()
    alphas: Sequence[Union[np.ndarray, float]] = ()


@attrs.define
class PageTextLineBoundingBoxStepOutput:
    score_maps: Sequence[ScoreMap] = ()
    colors: Sequence[Tuple[int, int, int]] = ()


@attrs.define
class PageTextLinePolygonCollection:
    height: int
    width: int
    polygons: Sequence[Polygon] = ()
    height_points_group_sizes: Sequence[int] = ()
    height_points_up: PointList = attrs.field(factory=PointList)
    height_points_down: PointList = attrs.field(factory=PointList)


@attrs.define
class PageCharPolygonCollection:
    height: int
    width: int
    char_polygons: Sequence[Polygon] = ()
    adjusted_char_polygons: Sequence[Polygon] = ()
    height_points_up: PointList = attrs.field(factory=PointList)
    height_points_down: PointList = attrs.field(factory=PointList)


@attrs.define
class PageTextLineLabelStepOutput:
    page_char_polygon_collection: PageCharPolygonCollection
    page_text_line_polygon_collection: PageTextLinePolygonCollection


# ---- the step ---------------------------------------------------------------------------------------------------
@attrs.define
class PageAssemblerStepConfig:
    pass


@attrs.define
class PageAssemblerStepInput:
    page_layout_step_output: PageLayoutStepOutput
    page_background_step_output: PageBackgroundStepOutput
    page_image_step_output: PageImageStepOutput
    page_barcode_step_output: PageBarcodeStepOutput
    page_text_line_step_output: PageTextLineStepOutput
    page_non_text_symbol_step_output: PageNonTextSymbolStepOutput
    page_text_line_bounding_box_step_output: PageTextLineBoundingBoxStepOutput
    page_text_line_label_step_output: PageTextLineLabelStepOutput


@attrs.define
class PageDisconnectedTextRegionCollection:
    disconnected_text_regions: Sequence[DisconnectedTextRegion]

    def to_polygons(self):
        for disconnected_text_region in self.disconnected_text_regions:
            yield disconnected_text_region.polygon


@attrs.define
class PageNonTextRegionCollection:
    non_text_regions: Sequence[NonTextRegion]

    def to_polygons(self):
        for non_text_region in self.non_text_regions:
            yield non_text_region.polygon


@attrs.define
class PageSealImpressionCharPolygonCollection:
    char_polygons: Sequence[Polygon]


@attrs.define
class Page(Shapable):
    image: Image
    page_image_collection: PageImageCollection
    page_bottom_layer_image: Image
    page_text_line_collection: PageTextLineCollection
    page_seal_impression_text_line_collection: PageSealImpressionTextLineCollection
    page_char_polygon_collection: PageCharPolygonCollection
    page_text_line_polygon_collection: PageTextLinePolygonCollection
    page_disconnected_text_region_collection: PageDisconnectedTextRegionCollection
    page_non_text_region_collection: PageNonTextRegionCollection
    page_seal_impression_char_polygon_collection: PageSealImpressionCharPolygonCollection

    @property
    def height(self):
        return self.image.height

    @property
    def width(self):
        return self.image.width


@attrs.define
class PageAssemblerStepOutput:
    page: Page


class PageAssemblerStep(PipelineStep[PageAssemblerStepConfig, PageAssemblerStepInput, PageAssemblerStepOutput]):

    def run(self, input: PageAssemblerStepInput, rng: RandomGenerator):
        page_layout = input.page_layout_step_output.page_layout
        background_image = input.page_background_step_output.background_image
        page_image_collection = input.page_image_step_output.page_image_collection
        page_bottom_layer_image = input.page_image_step_output.page_bottom_layer_image
        barcodes = input.page_barcode_step_output
        page_text_line_collection = input.page_text_line_step_output.page_text_line_collection
        seal_collection = input.page_text_line_step_output.page_seal_impression_text_line_collection
        symbols = input.page_non_text_symbol_step_output
        bounding_boxes = input.page_text_line_bounding_box_step_output
        labels = input.page_text_line_label_step_output

        assert background_image.mat.shape == (page_layout.height, page_layout.width, 3)
        # the page is assembled on the device: the background (the copy the reference makes, page_assembler.py:143) is the FIRST layer of the
        # one composite launch -- a plain full-page copy, read in place from the page-locked ring like every other layer plane, so the
        # page needs no upload of its own (a synchronous 3 MB copy: three runtime dispatches and a stream synchronisation per page) --
        # and the steps that follow (page distortion, resizing) take the page where it is: ``.mat`` downloads it on first touch
        ctx = _native.default_ctx()
        background_mat = background_image.mat
        as_layer = (isinstance(background_mat, np.ndarray) and background_mat.dtype == np.uint8 and background_mat.ndim == 3
                    and background_image.box is None)
        assembled_image = attrs.evolve(background_image, mat=ctx.dev_empty(background_mat.shape, np.uint8) if as_layer
                                       else ctx.to_device(background_mat))

        # Seal impressions are rotated first (device warps, independent of the page); their layers are recorded
        # last, so the composite order is untouched.
        seal_layers = []
        page_seal_impression_char_polygons: List[Polygon] = []
        for seal_impression, resource in zip(seal_collection.seal_impressions,
                                             seal_collection.seal_impression_resources):
            rotated = rotate.distort({'angle': resource.angle}, mask=seal_impression.background_mask,
                                     score_map=resource.text_line_filled_score_map,
                                     polygons=resource.char_polygons or None)
            background_mask, text_score_map = rotated.mask, rotated.score_map
            assert background_mask is not None and text_score_map is not None
            assert background_mask.shape == text_score_map.shape
            center = resource.box.get_center_point()
            up = center.y - background_mask.height // 2
            down = up + background_mask.height - 1
            left = center.x - background_mask.width // 2
            right = left + background_mask.width - 1
            if up < 0 or down >= assembled_image.height or left < 0 or right >= assembled_image.width:
                continue  # reference :214-217: the seal impression is simply dropped
            box = Box(up=up, down=down, left=left, right=right)
            seal_layers.append((box, seal_impression.color, background_mask, seal_impression.alpha, text_score_map))
            page_seal_impression_char_polygons.extend(
                polygon.to_shifted_polygon(offset_y=up, offset_x=left) for polygon in (rotated.polygons or ()))

        with deferred_fill(assembled_image.arr):
            if as_layer:
                Box(up=0, down=assembled_image.height - 1, left=0, right=assembled_image.width - 1).fill_image(
                    assembled_image, background_mat, alpha=1.0)
            for page_image in page_image_collection.page_images:
                page_image.box.fill_image(assembled_image, page_image.image, alpha=page_image.alpha)
            for score_map in barcodes.barcode_qr_score_maps:
                assembled_image[score_map] = (0, 0, 0)
            for score_map in barcodes.barcode_code39_score_maps:
                assembled_image[score_map] = (0, 0, 0)
            for score_map, color in zip(bounding_boxes.score_maps, bounding_boxes.colors):
                assembled_image[score_map] = color
            for text_line in page_text_line_collection.text_lines:
                if text_line.score_map:
                    text_line.score_map.fill_image(assembled_image, text_line.glyph_color)
                else:
                    text_line.mask.fill_image(assembled_image, text_line.image)
            for image, box, alpha in zip(symbols.images, symbols.boxes, symbols.alphas):
                box.fill_image(assembled_image, value=image, alpha=alpha)
            for box, color, background_mask, alpha, text_score_map in seal_layers:
                box.fill_image(assembled_image, value=color, image_mask=background_mask, alpha=alpha)
                box.fill_image(assembled_image, value=color, alpha=text_score_map)

        page = Page(
            image=assembled_image,
            page_image_collection=page_image_collection,
            page_bottom_layer_image=page_bottom_layer_image,
            page_text_line_collection=page_text_line_collection,
            page_seal_impression_text_line_collection=seal_collection,
            page_char_polygon_collection=labels.page_char_polygon_collection,
            page_text_line_polygon_collection=labels.page_text_line_polygon_collection,
            page_disconnected_text_region_collection=PageDisconnectedTextRegionCollection(
                page_layout.disconnected_text_regions),
            page_non_text_region_collection=PageNonTextRegionCollection(page_layout.non_text_regions),
            page_seal_impression_char_polygon_collection=PageSealImpressionCharPolygonCollection(
                char_polygons=page_seal_impression_char_polygons),
        )
        return PageAssemblerStepOutput(page=page)


page_assembler_step_factory = PipelineStepFactory(PageAssemblerStep)
