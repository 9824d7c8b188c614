"""Page distortion: the chain driver of the synthetic-page pipeline and the label rasterisation that follows it
(reference: vkit/pipeline/text_detection/page_distortion.py:128-473).

``run`` (reference :316-473): flatten the page's polygons / points, build the 1-px-border active mask, hand image +
mask + polygons + points to the ``RandomDistortion`` chain, fill the inactive (black) region from the bottom
layer image, unflatten, rasterise the labels.

MI355X shape: the image / mask pair goes through the shared-grid device pass of each geometric operator; the
labels (hundreds to thousands of polygons per page) are painted by ``vkx_paint_polys`` -- one ordered
ownership raster per label plane pair instead of one fillPoly + boolean-index assignment per polygon.  The
painter / debug image branches and the pluggable char-mask engines are outside the path (the default engine,
"every char polygon, keep max", is what is implemented; reference engine/char_mask/default.py:44-53).
"""
import itertools
from typing import Any, Generic, List, Mapping, Optional, Sequence, Tuple, TypeVar, Union

import attrs
import numpy as np
from numpy.random import Generator as RandomGenerator

from vkit_amd import _native
from vkit_amd.element import Image, Mask, Point, PointArray, PointList, Polygon, PolygonSoup, ScoreMap
from vkit_amd.mechanism.distortion_policy import RandomDistortionDebug, random_distortion_factory
from vkit_amd.utility import PathType
from ..interface import PipelineStep, PipelineStepFactory
from .page_assembler import (
    DisconnectedTextRegion,
    NonTextRegion,
    PageAssemblerStepOutput,
    PageCharPolygonCollection,
    PageDisconnectedTextRegionCollection,
    PageNonTextRegionCollection,
    PageSealImpressionCharPolygonCollection,
    PageTextLinePolygonCollection,
)


@attrs.define
class PageDistortionStepConfig:
    random_distortion_factory_config: Optional[Union[Mapping[str, Any], PathType]] = attrs.field(
        factory=lambda: {'disabled_policy_names': ['defocus_blur', 'zoom_in_blur']})
    enable_debug_random_distortion: bool = False
    enable_distorted_char_mask: bool = True
    enable_distorted_seal_impression_char_mask: bool = True
    char_mask_engine_config: Mapping[str, Any] = attrs.field(factory=lambda: {'type': 'default'})
    enable_distorted_char_height_score_map: bool = True
    enable_debug_distorted_char_heights: bool = False
    enable_distorted_text_line_mask: bool = True
    enable_distorted_text_line_height_score_map: bool = True
    enable_debug_distorted_text_line_heights: bool = False


@attrs.define
class PageDistortionStepInput:
    page_assembler_step_output: PageAssemblerStepOutput


@attrs.define
class PageDistortionStepOutput:
    page_image: Image
    page_random_distortion_debug: Optional[RandomDistortionDebug]
    page_active_mask: Mask
    page_char_polygon_collection: PageCharPolygonCollection
    page_char_mask: Optional[Mask]
    page_seal_impression_char_mask: Optional[Mask]
    page_char_height_score_map: Optional[ScoreMap]
    page_char_heights: Optional[Sequence[float]]
    page_char_heights_debug_image: Optional[Image]
    page_text_line_polygon_collection: PageTextLinePolygonCollection
    page_text_line_mask: Optional[Mask]
    page_text_line_height_score_map: Optional[ScoreMap]
    page_text_line_heights: Optional[Sequence[float]]
    page_text_line_heights_debug_image: Optional[Image]
    page_disconnected_text_region_collection: PageDisconnectedTextRegionCollection
    page_non_text_region_collection: PageNonTextRegionCollection
    page_seal_impression_char_polygon_collection: PageSealImpressionCharPolygonCollection


_E = TypeVar('_E', Point, Polygon)


class ElementFlattener(Generic[_E]):

    def __init__(self, grouped_elements: Sequence[Sequence[_E]]):
        self.grouped_elements = grouped_elements
        self.group_sizes = [len(elements) for elements in grouped_elements]

    def flatten(self):
        return tuple(itertools.chain.from_iterable(self.grouped_elements))

    def flatten_polygons(self) -> PolygonSoup:
        """The groups as ONE array-backed sequence of polygons (element/soup.py): the chain's geometric operators, its
        clipping and the label paint then work on the vertex array and build no ``Point`` / ``Polygon`` object."""
        if any(isinstance(group, PolygonSoup) for group in self.grouped_elements):
            return PolygonSoup.concatenate([PolygonSoup.from_polygons(group) for group in self.grouped_elements])
        return PolygonSoup.from_polygons(itertools.chain.from_iterable(self.grouped_elements))     # one pass, one concatenate

    def flatten_points(self) -> PointArray:
        if any(isinstance(group, PointArray) for group in self.grouped_elements):
            groups = [PointArray.from_points(group) for group in self.grouped_elements]
            return PointArray(np.concatenate([g.smooth_xy for g in groups], axis=0) if groups else np.zeros((0, 2)))
        flat = PointArray.from_points(itertools.chain.from_iterable(self.grouped_elements))
        return PointArray(flat.smooth_xy)

    def unflatten(self, flattened_elements: Sequence[_E]) -> Sequence[Sequence[_E]]:
        assert len(flattened_elements) == sum(self.group_sizes)
        grouped, begin = [], 0
        for group_size in self.group_sizes:
            grouped.append(flattened_elements[begin:begin + group_size])
            begin += group_size
        return grouped


class _PaintQueue:
    """The label paints of one page, gathered: every ``paint_polygons`` call of a page writes fresh planes of the page's shape, so the
    plane sets are recorded (their device planes exist, uninitialised, and the elements that wrap them are handed out) and painted by
    ONE ``vkx_paint_poly_sets_fresh_dev`` call when the page's labels are complete -- four sets per page: one table copy and three
    kernels instead of four times that."""

    def __init__(self, shape):
        self.shape = tuple(shape)
        self.sets = []

    def add(self, polygons, values, want_mask):
        ctx = _native.default_ctx()
        mask = ctx.dev_empty(self.shape, np.uint8) if want_mask else None
        score = ctx.dev_empty(self.shape, np.float32) if values is not None else None
        if isinstance(polygons, PolygonSoup):
            flat, offsets = polygons.int_xy, polygons.offsets
        else:
            pts = [np.asarray(polygon.to_np_array(), dtype=np.int32).reshape(-1, 2) for polygon in polygons]
            offsets = np.zeros(len(pts) + 1, np.int32)
            if pts:
                offsets[1:] = np.cumsum([len(p) for p in pts])
            flat = np.concatenate(pts, axis=0) if pts else np.zeros((0, 2), np.int32)
        self.sets.append((flat, offsets, values, mask, score))
        return mask, score

    def flush(self):
        sets, self.sets = self.sets, []
        for k in range(0, len(sets), 8):
            _native.paint_poly_sets_fresh(sets[k:k + 8], self.shape)


def paint_polygons(shape: Tuple[int, int], polygons: Sequence[Polygon], values: Optional[Sequence[float]] = None,
                   want_mask: bool = True, queue: Optional[_PaintQueue] = None):
    """Sequential ``polygon.fill_mask(mask)`` / ``polygon.fill_score_map(score_map, value)`` over ``polygons`` on
    fresh planes, as one ordered device paint.  Returns (Mask | None, ScoreMap | None).  With a ``queue`` (device-resident pages) the
    paint is recorded and runs with the page's other label paints (``_PaintQueue.flush``)."""
    height, width = shape
    resident = _native.resident_mode()
    if resident and queue is not None and queue.shape == (height, width) and (want_mask or values is not None):
        np_mask, np_score = queue.add(polygons, values, want_mask)
        return (Mask(mat=np_mask) if want_mask else None,
                ScoreMap(mat=np_score, is_prob=False) if values is not None else None)
    if resident and len(polygons):
        # the planes are painted where the page lives and stay there until somebody reads ``.mat``; the paint writes every pixel of
        # them (0 outside every polygon): no memset per plane
        ctx = _native.default_ctx()
        np_mask = ctx.dev_empty((height, width), np.uint8) if want_mask else None
        np_score = ctx.dev_empty((height, width), np.float32) if values is not None else None
    elif resident:
        np_mask = _native.dev_zeros((height, width), np.uint8) if want_mask else None
        np_score = _native.dev_zeros((height, width), np.float32) if values is not None else None
    else:
        np_mask = np.zeros((height, width), np.uint8) if want_mask else None
        np_score = np.zeros((height, width), np.float32) if values is not None else None
    if len(polygons):
        # a polygon's raster is defined on its bounding box with the integer vertices made box-relative (reference
        # element/polygon.py:105-138,70-77): shifting them back by the integer box origin gives the integer vertices
        # themselves, so the polygons go to the device as they are (no per-polygon box / relative-polygon objects)
        if isinstance(polygons, PolygonSoup):
            _native.paint_polys_flat(polygons.int_xy, polygons.offsets, values=values, mask=np_mask, score=np_score, fresh=resident)
        else:
            _native.paint_polys([polygon.to_np_array() for polygon in polygons], values=values, mask=np_mask, score=np_score, fresh=resident)
    mask = Mask(mat=np_mask) if want_mask else None
    score_map = ScoreMap(mat=np_score, is_prob=False) if values is not None else None
    return mask, score_map


def _heights(points_up: PointList, points_down: PointList):
    np_heights = np.linalg.norm(points_down.to_smooth_np_array() - points_up.to_smooth_np_array(), axis=1)
    np_heights += 1  # "Add one to compensate." (reference :186, :268)
    return np_heights


def _group_means(values: np.ndarray, group_sizes: Sequence[int]) -> List[float]:
    """``values[begin:begin + size].mean()`` per group (reference page_distortion.py:188-196), for all groups at once when every
    group has fewer than 8 elements: ``.mean()`` then adds its slice one element after the other starting from 0 (numpy's pairwise
    sum starts at 8 elements), so the groups are laid out as rows of a zero-padded matrix and the columns added in order -- the same
    additions (x + 0.0 is x), then the same division by the count; tests/test_soup.py compares the two forms.  Larger or empty groups
    take the slice form.  (``np.add.reduceat`` associates differently: first element + sum of the rest.)"""
    sizes = np.asarray(group_sizes, dtype=np.int64)
    if sizes.size == 0:
        return []
    if (sizes <= 0).any() or (sizes >= 8).any() or values.dtype not in (np.float32, np.float64):
        out, begin = [], 0
        for size in group_sizes:
            out.append(float(values[begin:begin + size].mean()))
            begin += size
        return out
    starts = np.cumsum(sizes) - sizes
    width = int(sizes.max())
    cols = np.arange(width)
    padded = np.where(cols[None, :] < sizes[:, None], values[np.minimum(starts[:, None] + cols[None, :], values.shape[0] - 1)],
                      values.dtype.type(0))
    acc = np.zeros(sizes.shape[0], values.dtype)
    for c in range(width):
        acc = acc + padded[:, c]
    return [float(v) for v in (acc / sizes.astype(values.dtype))]


class PageDistortionStep(PipelineStep[PageDistortionStepConfig, PageDistortionStepInput, PageDistortionStepOutput]):

    def __init__(self, config: PageDistortionStepConfig):
        super().__init__(config)
        self.random_distortion = random_distortion_factory.create(self.config.random_distortion_factory_config)
        engine_type = dict(self.config.char_mask_engine_config).get('type', 'default')
        if engine_type != 'default':
            raise NotImplementedError(f'char mask engine "{engine_type}" is outside the accelerated path')
        for flag in ('enable_debug_distorted_char_heights', 'enable_debug_distorted_text_line_heights'):
            if getattr(self.config, flag):
                raise NotImplementedError(f'{flag}: the painter is outside the accelerated path')

    @classmethod
    def fill_page_inactive_region(cls, page_image: Image, page_active_mask: Mask, page_bottom_layer_image: Image):
        assert page_image.shape == page_active_mask.shape
        if page_bottom_layer_image.shape != page_image.shape:
            page_bottom_layer_image = page_bottom_layer_image.to_resized_image(
                resized_height=page_image.height, resized_width=page_image.width)
        if page_image.on_device and page_image.arr.ndim == 3:
            # the same fill where the page lives: inverted mask (a table look-up) selects, the bottom layer is the value
            ctx = page_image.arr.ctx
            inverted = ctx.to_device(page_active_mask.to_inverted_mask().arr)
            # (a bottom layer still on the host is read by the fill where it is: staged in the library's page-locked ring, no upload)
            layer = _native.make_layer((0, 0, page_image.height, page_image.width), page_image.arr.shape[2],
                                       page_bottom_layer_image.arr, mask=inverted)
            _native.fill(page_image.arr, [layer])
            return
        page_active_mask.to_inverted_mask().fill_image(page_image, page_bottom_layer_image)

    def generate_text_line_labelings(self, distorted_image: Image, text_line_polygons: Sequence[Polygon],
                                     text_line_height_points_up: PointList,
                                     text_line_height_points_down: PointList,
                                     text_line_height_points_group_sizes: Sequence[int], _paint_queue: Optional[_PaintQueue] = None):
        text_line_heights: Optional[List[float]] = None
        if self.config.enable_distorted_text_line_height_score_map:
            np_heights = _heights(text_line_height_points_up, text_line_height_points_down)
            assert sum(text_line_height_points_group_sizes) == np_heights.shape[0]
            text_line_heights = _group_means(np_heights, text_line_height_points_group_sizes)
        text_line_mask, text_line_height_score_map = None, None
        if self.config.enable_distorted_text_line_mask or text_line_heights is not None:
            text_line_mask, text_line_height_score_map = paint_polygons(
                distorted_image.shape, text_line_polygons, values=text_line_heights,
                want_mask=self.config.enable_distorted_text_line_mask, queue=_paint_queue)
        return text_line_mask, text_line_height_score_map, text_line_heights, None

    def generate_char_labelings(self, distorted_image: Image, char_polygons: Sequence[Polygon],
                                seal_impression_char_polygons: Sequence[Polygon],
                                char_height_points_up: PointList, char_height_points_down: PointList,
                                _paint_queue: Optional[_PaintQueue] = None):
        char_mask: Optional[Mask] = None
        if self.config.enable_distorted_char_mask:
            # default engine: every polygon.fill_mask(mask, keep_max_value=True) with value 1 == union
            char_mask, _ = paint_polygons(distorted_image.shape, char_polygons, queue=_paint_queue)
        seal_impression_char_mask: Optional[Mask] = None
        if self.config.enable_distorted_seal_impression_char_mask:
            seal_impression_char_mask, _ = paint_polygons(distorted_image.shape, seal_impression_char_polygons, queue=_paint_queue)

        char_height_score_map: Optional[ScoreMap] = None
        char_heights: Optional[List[float]] = None
        if self.config.enable_distorted_char_height_score_map:
            np_heights = _heights(char_height_points_up, char_height_points_down)
            # Large heights first, so that the small height survives where two char boxes overlap (reference :270-273)
            order: Tuple[int, ...] = tuple(reversed(np_heights.argsort()))
            char_heights = [0.0] * len(char_polygons)
            for idx in order:
                char_heights[idx] = float(np_heights[idx])
            ordered = (char_polygons.reordered(order) if isinstance(char_polygons, PolygonSoup)
                       else [char_polygons[idx] for idx in order])
            _, char_height_score_map = paint_polygons(
                distorted_image.shape, ordered, values=[char_heights[idx] for idx in order], want_mask=False, queue=_paint_queue)
        return char_mask, seal_impression_char_mask, char_height_score_map, char_heights, None

    def run(self, input: PageDistortionStepInput, rng: RandomGenerator):
        page = input.page_assembler_step_output.page
        page_char_polygon_collection = page.page_char_polygon_collection
        page_text_line_polygon_collection = page.page_text_line_polygon_collection
        page_disconnected_text_region_collection = page.page_disconnected_text_region_collection
        page_non_text_region_collection = page.page_non_text_region_collection

        polygon_flattener = ElementFlattener([
            page_char_polygon_collection.char_polygons,
            page_char_polygon_collection.adjusted_char_polygons,
            page_text_line_polygon_collection.polygons,
            tuple(page_disconnected_text_region_collection.to_polygons()),
            tuple(page_non_text_region_collection.to_polygons()),
            page.page_seal_impression_char_polygon_collection.char_polygons,
        ])
        point_flattener = ElementFlattener([
            page_char_polygon_collection.height_points_up,
            page_char_polygon_collection.height_points_down,
            page_text_line_polygon_collection.height_points_up,
            page_text_line_polygon_collection.height_points_down,
        ])

        page_random_distortion_debug = RandomDistortionDebug() if self.config.enable_debug_random_distortion else None

        # 1-px border off: mitigates cv.remap's border interpolation (reference :357-364)
        height, width = page.image.shape
        if page.image.on_device and height >= 3 and width >= 3:
            # the same mask painted where the page lives (one launch: ones, then the four border strips) instead of a host plane
            # that the first geometric operator would have to upload
            ctx = page.image.arr.ctx
            plane = ctx.dev_empty((height, width), np.uint8)
            _native.fill(plane, [_native.make_layer(box, 1, value) for box, value in (
                ((0, 0, height, width), 1), ((0, 0, 1, width), 0), ((height - 1, 0, 1, width), 0), ((0, 0, height, 1), 0),
                ((0, width - 1, height, 1), 0))])
            page_active_mask = Mask(mat=plane)
        else:
            page_active_mask = Mask.from_shapable(page.image, value=1)
            with page_active_mask.writable_context:
                page_active_mask.mat[0] = 0
                page_active_mask.mat[-1] = 0
                page_active_mask.mat[:, 0] = 0
                page_active_mask.mat[:, -1] = 0

        # everything from here to the end of the step stays on the device (element ``.mat`` downloads on first touch):
        # the operators of the chain hand DevArrays to each other, the inactive-region fill and the label paint run where
        # the page is, and a PageResizingStep behind this one reads the planes from HBM
        with _native.resident():
            return self._run_resident(page, rng, page_active_mask, polygon_flattener, point_flattener,
                                      page_random_distortion_debug, page_char_polygon_collection,
                                      page_text_line_polygon_collection)

    def _run_resident(self, page, rng, page_active_mask, polygon_flattener, point_flattener, page_random_distortion_debug,
                      page_char_polygon_collection, page_text_line_polygon_collection):
        result = self.random_distortion.distort(
            image=page.image,
            mask=page_active_mask,
            polygons=polygon_flattener.flatten_polygons(),
            points=point_flattener.flatten_points(),
            rng=rng,
            debug=page_random_distortion_debug,
        )
        assert result.image and result.mask
        polygons = result.polygons or ()
        points = result.points or ()

        self.fill_page_inactive_region(page_image=result.image, page_active_mask=result.mask,
                                       page_bottom_layer_image=page.page_bottom_layer_image)

        (char_polygons, adjusted_char_polygons, text_line_polygons, disconnected_text_region_polygons,
         non_text_region_polygons, seal_impression_char_polygons) = polygon_flattener.unflatten(polygons)
        (char_height_points_up, char_height_points_down, text_line_height_points_up,
         text_line_height_points_down) = (
            group.to_point_list() if isinstance(group, PointArray) else PointList(group)
            for group in point_flattener.unflatten(points))

        text_line_height_points_group_sizes = page_text_line_polygon_collection.height_points_group_sizes
        assert len(text_line_polygons) == len(text_line_height_points_group_sizes)
        assert len(text_line_height_points_up) == len(text_line_height_points_down)

        paint_queue = _PaintQueue(result.image.shape)          # the page's label paints as one call
        (text_line_mask, text_line_height_score_map, text_line_heights,
         text_line_heights_debug_image) = self.generate_text_line_labelings(
            _paint_queue=paint_queue,
            distorted_image=result.image,
            text_line_polygons=text_line_polygons,
            text_line_height_points_up=text_line_height_points_up,
            text_line_height_points_down=text_line_height_points_down,
            text_line_height_points_group_sizes=text_line_height_points_group_sizes,
        )
        (char_mask, seal_impression_char_mask, char_height_score_map, char_heights,
         char_heights_debug_image) = self.generate_char_labelings(
            _paint_queue=paint_queue,
            distorted_image=result.image,
            char_polygons=char_polygons,
            seal_impression_char_polygons=seal_impression_char_polygons,
            char_height_points_up=char_height_points_up,
            char_height_points_down=char_height_points_down,
        )

        paint_queue.flush()
        height, width = result.image.height, result.image.width
        return PageDistortionStepOutput(
            page_image=result.image,
            page_random_distortion_debug=page_random_distortion_debug,
            page_active_mask=result.mask,
            page_char_polygon_collection=PageCharPolygonCollection(
                height=height, width=width, char_polygons=char_polygons,
                adjusted_char_polygons=adjusted_char_polygons, height_points_up=char_height_points_up,
                height_points_down=char_height_points_down),
            page_char_mask=char_mask,
            page_seal_impression_char_mask=seal_impression_char_mask,
            page_char_height_score_map=char_height_score_map,
            page_char_heights=char_heights,
            page_char_heights_debug_image=char_heights_debug_image,
            page_text_line_polygon_collection=PageTextLinePolygonCollection(
                height=height, width=width, polygons=text_line_polygons,
                height_points_group_sizes=text_line_height_points_group_sizes,
                height_points_up=text_line_height_points_up, height_points_down=text_line_height_points_down),
            page_text_line_mask=text_line_mask,
            page_text_line_height_score_map=text_line_height_score_map,
            page_text_line_heights=text_line_heights,
            page_text_line_heights_debug_image=text_line_heights_debug_image,
            page_disconnected_text_region_collection=PageDisconnectedTextRegionCollection(
                disconnected_text_regions=[DisconnectedTextRegion(polygon=polygon)
                                           for polygon in disconnected_text_region_polygons]),
            page_non_text_region_collection=PageNonTextRegionCollection(
                non_text_regions=[NonTextRegion(polygon=polygon) for polygon in non_text_region_polygons]),
            page_seal_impression_char_polygon_collection=PageSealImpressionCharPolygonCollection(
                char_polygons=seal_impression_char_polygons),
        )


page_distortion_step_factory = PipelineStepFactory(PageDistortionStep)
