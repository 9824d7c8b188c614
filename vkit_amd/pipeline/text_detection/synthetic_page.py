"""A synthetic, seeded input for ``PageAssemblerStep`` shaped like BASELINE config 4 (a 1024^2 page, gray background, text-line
layers with float32 alpha, then the distortion chain): the workload of ``bench.py --config c4``, ``tools/pool_scale.py`` and the page
tests.  Data only -- every value comes from ``default_rng(seed)``; nothing here is reference code (the reference builds its pages from
fonts and corpora, vkit/pipeline/text_detection/page_text_line.py, out of scope)."""
import numpy as np
from numpy.random import default_rng


def synthetic_page_input(seed, size=256, n_lines=24):
    """A C4-shaped page: gray background, one page image, text-line score-map layers, one symbol, one seal."""
    from vkit_amd.element import Box, Image, Mask, Point, PointList, Polygon, ScoreMap
    from vkit_amd.pipeline import text_detection as T
    rng = default_rng(seed)
    gray = int(rng.integers(127, 256))
    background = Image(mat=np.full((size, size, 3), gray, np.uint8))
    bottom = Image(mat=rng.integers(0, 256, (size, size, 3), dtype=np.uint8))
    page_images = [T.PageImage(image=Image(mat=rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)),
                               box=Box(up=10, down=49, left=20, right=79), alpha=0.8)]
    text_lines, polygons, ups, downs, sizes, char_polygons, char_ups, char_downs = [], [], [], [], [], [], [], []
    lh, lw = size // 16, size // 2
    for i in range(n_lines):
        up = int(rng.integers(0, size - lh))
        left = int(rng.integers(0, size - lw))
        box = Box(up=up, down=up + lh - 1, left=left, right=left + lw - 1)
        alpha = (rng.random((lh, lw), dtype=np.float32) * (rng.random((lh, lw)) < 0.3)).astype(np.float32)
        if i % 5 == 4:  # a line without score map: mask + rendered image
            text_lines.append(T.TextLine(image=Image(mat=rng.integers(0, 256, (lh, lw, 3), dtype=np.uint8), box=box),
                                         mask=Mask(mat=(alpha > 0).astype(np.uint8), box=box), score_map=None,
                                         glyph_color=(10, 20, 30)))
        else:
            text_lines.append(T.TextLine(image=Image(mat=np.zeros((lh, lw, 3), np.uint8), box=box),
                                         mask=Mask(mat=(alpha > 0).astype(np.uint8), box=box),
                                         score_map=ScoreMap(mat=alpha, box=box), glyph_color=(10, 20, 30)))
        polygons.append(Polygon.from_xy_pairs([(left, up), (left + lw - 1, up), (left + lw - 1, up + lh - 1),
                                               (left, up + lh - 1)]))
        ups.extend([Point.create(y=up, x=left), Point.create(y=up, x=left + lw - 1)])
        downs.extend([Point.create(y=up + lh - 1, x=left), Point.create(y=up + lh - 1, x=left + lw - 1)])
        sizes.append(2)
        for c in range(6):
            cl = left + c * (lw // 6)
            char_polygons.append(Polygon.from_xy_pairs([(cl, up), (cl + lw // 6 - 2, up), (cl + lw // 6 - 2, up + lh - 1),
                                                        (cl, up + lh - 1)]))
            char_ups.append(Point.create(y=up, x=cl + 3))
            char_downs.append(Point.create(y=up + lh - 1 - (c % 3), x=cl + 3))
    seal_mask = np.zeros((41, 41), np.uint8)
    yy, xx = np.ogrid[:41, :41]
    seal_mask[((yy - 20) ** 2 + (xx - 20) ** 2 <= 400) & ((yy - 20) ** 2 + (xx - 20) ** 2 >= 300)] = 1
    seal_text = (rng.random((41, 41), dtype=np.float32) * (rng.random((41, 41)) < 0.2)).astype(np.float32)
    seals = T.PageSealImpressionTextLineCollection(
        height=size, width=size,
        seal_impressions=[T.SealImpression(alpha=0.7, color=(200, 20, 30), background_mask=Mask(mat=seal_mask))],
        seal_impression_resources=[T.SealImpressionResource(
            box=Box(up=size // 2, down=size // 2 + 40, left=size // 2, right=size // 2 + 40), angle=25,
            text_line_filled_score_map=ScoreMap(mat=seal_text),
            char_polygons=[Polygon.from_xy_pairs([(5, 5), (15, 5), (15, 15), (5, 15)])])])
    barcode = np.zeros((size, size), np.float32)
    barcode[size - 40:size - 10, 10:70:2] = 1.0
    bbox_alpha = np.zeros((size, size), np.float32)
    bbox_alpha[5:8, 5:size - 5] = 0.5
    return T.PageAssemblerStepInput(
        page_layout_step_output=T.PageLayoutStepOutput(T.PageLayout(
            height=size, width=size,
            disconnected_text_regions=[T.DisconnectedTextRegion(polygons[0])],
            non_text_regions=[T.NonTextRegion(Polygon.from_xy_pairs([(3, 3), (30, 4), (28, 28), (4, 30)]))])),
        page_background_step_output=T.PageBackgroundStepOutput(background),
        page_image_step_output=T.PageImageStepOutput(
            page_image_collection=T.PageImageCollection(height=size, width=size, page_images=page_images),
            page_bottom_layer_image=bottom),
        page_barcode_step_output=T.PageBarcodeStepOutput(height=size, width=size,
                                                         barcode_qr_score_maps=[ScoreMap(mat=barcode)]),
        page_text_line_step_output=T.PageTextLineStepOutput(
            page_text_line_collection=T.PageTextLineCollection(height=size, width=size, text_lines=text_lines),
            page_seal_impression_text_line_collection=seals),
        page_non_text_symbol_step_output=T.PageNonTextSymbolStepOutput(
            images=[Image(mat=rng.integers(0, 256, (12, 12, 3), dtype=np.uint8))],
            boxes=[Box(up=100, down=111, left=7, right=18)],
            alphas=[(rng.random((12, 12), dtype=np.float32)).astype(np.float32)]),
        page_text_line_bounding_box_step_output=T.PageTextLineBoundingBoxStepOutput(
            score_maps=[ScoreMap(mat=bbox_alpha)], colors=[(255, 0, 0)]),
        page_text_line_label_step_output=T.PageTextLineLabelStepOutput(
            page_char_polygon_collection=T.PageCharPolygonCollection(
                height=size, width=size, char_polygons=char_polygons, adjusted_char_polygons=char_polygons,
                height_points_up=PointList(char_ups), height_points_down=PointList(char_downs)),
            page_text_line_polygon_collection=T.PageTextLinePolygonCollection(
                height=size, width=size, polygons=polygons, height_points_group_sizes=sizes,
                height_points_up=PointList(ups), height_points_down=PointList(downs))),
    )
