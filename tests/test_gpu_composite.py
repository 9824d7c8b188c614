"""Composite, label rasterisation, bicubic resize and the two pipeline steps, GPU (through the C ABI) vs oracle."""
import os

import numpy as np
import pytest
from numpy.random import default_rng

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def N():
    from vkit_amd import _native
    return _native


# ------------------------------------------------------------------ fill modes / float32 destinations
def test_fill_modes_against_reference_goldens(N, golden_dir):
    from vkit_amd.element import Box, Mask, ScoreMap
    F = np.load(os.path.join(golden_dir, 'fill_modes.npz'))
    sm0, m, val = F['f32_in'], F['f32_mask'], F['f32_value']
    for tag, kwargs in (('plain', {}), ('max', dict(keep_max_value=True)), ('min', dict(keep_min_value=True))):
        sm = ScoreMap(mat=sm0.copy(), is_prob=False)
        Mask(mat=m).fill_score_map(sm, 12.5, **kwargs)
        np.testing.assert_array_equal(sm.mat, F[f'f32_mask_const_{tag}'])
        sm = ScoreMap(mat=sm0.copy(), is_prob=False)
        Mask(mat=m).fill_score_map(sm, ScoreMap(mat=val, is_prob=False), **kwargs)
        np.testing.assert_array_equal(sm.mat, F[f'f32_mask_plane_{tag}'])
        sm = ScoreMap(mat=sm0.copy(), is_prob=False)
        Box(up=2, down=19, left=5, right=33).fill_score_map(sm, 7.25, **kwargs)
        np.testing.assert_array_equal(sm.mat, F[f'f32_box_const_{tag}'])
    mk0, mv = F['u8_in'], F['u8_value']
    for tag, kwargs in (('max', dict(keep_max_value=True)), ('min', dict(keep_min_value=True))):
        mk = Mask(mat=mk0.copy())
        Mask(mat=m).fill_mask(mk, 2, **kwargs)
        np.testing.assert_array_equal(mk.mat, F[f'u8_mask_const_{tag}'])
        mk = Mask(mat=mk0.copy())
        Mask(mat=m).fill_mask(mk, mv, **kwargs)
        np.testing.assert_array_equal(mk.mat, F[f'u8_mask_plane_{tag}'])
    from vkit_amd.element.opt import fill_np_array
    dst = sm0.copy()
    fill_np_array(dst, val, np_mask=F['f32_alpha'] > 0, alpha=F['f32_alpha'])
    np.testing.assert_array_equal(dst, F['f32_alpha_plane'])
    dst = sm0.copy()
    fill_np_array(dst, 3.0, alpha=0.3)
    np.testing.assert_array_equal(dst, F['f32_alpha_scalar'])
    # the reference fails on (2-D destination, mask, fractional scalar alpha): same exception type
    with pytest.raises(IndexError):
        fill_np_array(sm0.copy(), 3.0, np_mask=m.astype(bool), alpha=0.3)


def test_deferred_fill_equals_sequential(N):
    from vkit_amd.element import Box, Image, Mask, ScoreMap
    from vkit_amd.element.opt import deferred_fill
    rng = default_rng(11)
    base = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    layers = []
    for _ in range(40):
        h, w = int(rng.integers(4, 40)), int(rng.integers(4, 80))
        up, left = int(rng.integers(0, 120 - h)), int(rng.integers(0, 160 - w))
        kind = int(rng.integers(0, 4))
        box = Box(up=up, down=up + h - 1, left=left, right=left + w - 1)
        if kind == 0:
            alpha = (rng.random((h, w), dtype=np.float32) * (rng.random((h, w)) < 0.4)).astype(np.float32)
            layers.append(('score', ScoreMap(mat=alpha, box=box), tuple(int(v) for v in rng.integers(0, 256, 3))))
        elif kind == 1:
            layers.append(('box', box, Image(mat=rng.integers(0, 256, (h, w, 3), dtype=np.uint8)),
                           float(rng.random())))
        elif kind == 2:
            layers.append(('mask', Mask(mat=(rng.random((h, w)) < 0.5).astype(np.uint8), box=box),
                           Image(mat=rng.integers(0, 256, (h, w, 3), dtype=np.uint8))))
        else:
            layers.append(('box', box, Image(mat=rng.integers(0, 256, (h, w, 3), dtype=np.uint8)), 1.0))

    def apply(image):
        for layer in layers:
            if layer[0] == 'score':
                image[layer[1]] = layer[2]
            elif layer[0] == 'box':
                layer[1].fill_image(image, layer[2], alpha=layer[3])
            else:
                layer[1].fill_image(image, layer[2])

    sequential = Image(mat=base.copy())
    apply(sequential)
    deferred = Image(mat=base.copy())
    with deferred_fill(deferred.mat) as session:
        apply(deferred)
        assert len(session.layers) == len(layers)
        np.testing.assert_array_equal(deferred.mat, base)  # nothing written yet
    np.testing.assert_array_equal(deferred.mat, sequential.mat)
    # and both equal the oracle applied layer by layer
    want = base.copy()
    for layer in layers:
        box = layer[1].box if layer[0] != 'box' else layer[1]
        geo = (box.up, box.left, box.height, box.width)
        if layer[0] == 'score':
            O.fill(want, geo, layer[2], alpha=layer[1].mat)
        elif layer[0] == 'box':
            O.fill(want, geo, layer[2].mat, alpha=layer[3])
        else:
            O.fill(want, geo, layer[2].mat, mask=layer[1].mat)
    np.testing.assert_array_equal(sequential.mat, want)


def test_fills_on_a_device_resident_page(N):
    """Box / Mask / ScoreMap fills aimed at an Image whose pixels live in device memory (what PageAssemblerStep builds its
    page on): applied one by one, and recorded by a deferred composite and staged in one transfer
    (vkx_fill_u8_dev_host_layers) -- both equal the host-array result, and the page is downloaded only when ``.mat`` is read."""
    from vkit_amd.element import Box, Image, Mask, ScoreMap
    from vkit_amd.element.opt import deferred_fill
    rng = default_rng(12)
    base = rng.integers(0, 256, (150, 200, 3), dtype=np.uint8)
    layers = []
    for k in range(30):
        h, w = int(rng.integers(4, 50)), int(rng.integers(4, 90))
        up, left = int(rng.integers(0, 150 - h)), int(rng.integers(0, 200 - w))
        box = Box(up=up, down=up + h - 1, left=left, right=left + w - 1)
        if k % 3 == 0:
            alpha = (rng.random((h, w), dtype=np.float32) * (rng.random((h, w)) < 0.5)).astype(np.float32)
            layers.append(('score', ScoreMap(mat=alpha, box=box), tuple(int(v) for v in rng.integers(0, 256, 3))))
        elif k % 3 == 1:
            layers.append(('box', box, Image(mat=rng.integers(0, 256, (h, w, 3), dtype=np.uint8)), float(rng.random())))
        else:
            layers.append(('mask', Mask(mat=(rng.random((h, w)) < 0.5).astype(np.uint8), box=box),
                           Image(mat=rng.integers(0, 256, (h, w, 3), dtype=np.uint8))))

    def apply(image):
        for layer in layers:
            if layer[0] == 'score':
                image[layer[1]] = layer[2]
            elif layer[0] == 'box':
                layer[1].fill_image(image, layer[2], alpha=layer[3])
            else:
                layer[1].fill_image(image, layer[2])

    on_host = Image(mat=base.copy())
    apply(on_host)
    ctx = N.default_ctx()
    one_by_one = Image(mat=ctx.to_device(base))
    apply(one_by_one)
    assert one_by_one.on_device
    np.testing.assert_array_equal(one_by_one.mat, on_host.mat)
    deferred = Image(mat=ctx.to_device(base))
    with deferred_fill(deferred.arr) as session:
        apply(deferred)
        assert len(session.layers) == len(layers)
    assert deferred.on_device
    np.testing.assert_array_equal(deferred.mat, on_host.mat)


def test_deferred_fill_float32_layers(N):
    """A list of float32 layers (height-map style fills with masks, keep-max / keep-min, alpha planes) in one launch."""
    from vkit_amd.element import Box, Mask, ScoreMap
    from vkit_amd.element.opt import deferred_fill
    rng = default_rng(31)
    base = (rng.random((90, 130), dtype=np.float32) * 20).astype(np.float32)
    score = ScoreMap(mat=base.copy(), is_prob=False)
    want = base.copy()
    with deferred_fill(score.mat) as session:
        for i in range(30):
            h, w = int(rng.integers(3, 40)), int(rng.integers(3, 60))
            up, left = int(rng.integers(0, 90 - h)), int(rng.integers(0, 130 - w))
            box = Box(up=up, down=up + h - 1, left=left, right=left + w - 1)
            mask = (rng.random((h, w)) < 0.6).astype(np.uint8)
            value = float(rng.uniform(0, 40))
            mode = i % 3
            Mask(mat=mask, box=box).fill_score_map(score, value, keep_max_value=(mode == 1), keep_min_value=(mode == 2))
            O.fill(want, (up, left, h, w), value, mask=mask, mode=mode)
        assert len(session.layers) == 30
    np.testing.assert_array_equal(score.mat, want)


# ------------------------------------------------------------------ bicubic resize
@pytest.mark.parametrize('src_shape,dst_shape', [((20, 30), (33, 47)), ((64, 64), (64, 64)), ((97, 131), (40, 55)),
                                                 ((5, 7), (50, 3)), ((1, 1), (9, 9)), ((300, 200), (301, 199))])
def test_resize_cubic_small(N, src_shape, dst_shape):
    rng = default_rng(3)
    for cn in (1, 3, 4):
        shape = src_shape if cn == 1 else src_shape + (cn,)
        src = rng.integers(0, 256, shape, dtype=np.uint8)
        np.testing.assert_array_equal(N.resize_cubic(src, dst_shape), O.resize_cubic(src, dst_shape))
    f = (rng.random(src_shape, dtype=np.float32) * 40 - 5).astype(np.float32)
    np.testing.assert_array_equal(N.resize_cubic(f, dst_shape), O.resize_cubic(f, dst_shape))


def test_resize_cubic_extremes_and_elements(N):
    from vkit_amd.element import Image, Mask, ScoreMap
    rng = default_rng(4)
    # saturating overshoot: checkerboard of 0 / 255
    src = ((np.indices((40, 40)).sum(axis=0) % 2) * 255).astype(np.uint8)
    np.testing.assert_array_equal(N.resize_cubic(src, (93, 71)), O.resize_cubic(src, (93, 71)))
    image = Image(mat=rng.integers(0, 256, (60, 80, 3), dtype=np.uint8))
    np.testing.assert_array_equal(image.to_resized_image(resized_height=45).mat,
                                  O.resize_cubic(image.mat, (45, 60)))
    mask = Mask(mat=(rng.random((60, 80)) < 0.5).astype(np.uint8))
    want = (O.resize_cubic(mask.mat * 255, (90, 100)) > 0).astype(np.uint8)
    np.testing.assert_array_equal(mask.to_resized_mask(resized_height=90, resized_width=100).mat, want)
    score = ScoreMap(mat=rng.random((60, 80), dtype=np.float32))
    want = np.clip(O.resize_cubic(score.mat, (31, 99)), 0.0, 1.0)
    np.testing.assert_array_equal(score.to_resized_score_map(resized_height=31, resized_width=99).mat, want)


def test_resize_cubic_full_size_page(N):
    src = default_rng(5).integers(0, 256, (2048, 2048, 3), dtype=np.uint8)
    np.testing.assert_array_equal(N.resize_cubic(src, (2147, 2115)), O.resize_cubic(src, (2147, 2115)))


# ------------------------------------------------------------------ ordered polygon paint (label rasterisation)
def _paint_reference(shape, polygons, values):
    mask = np.zeros(shape, np.uint8)
    score = np.zeros(shape, np.float32)
    h, w = shape
    for pts, value in zip(polygons, values):
        pts = np.asarray(pts, np.int32)
        x0, y0 = int(pts[:, 0].min()), int(pts[:, 1].min())
        x1, y1 = int(pts[:, 0].max()), int(pts[:, 1].max())
        raster = O.fill_poly((y1 - y0 + 1, x1 - x0 + 1), pts - np.asarray([x0, y0], np.int32)).astype(bool)
        ys, xs = np.nonzero(raster)
        ys, xs = ys + y0, xs + x0
        keep = (ys >= 0) & (ys < h) & (xs >= 0) & (xs < w)
        mask[ys[keep], xs[keep]] = 1
        score[ys[keep], xs[keep]] = np.float32(value)
    return mask, score


def test_paint_polys_overlaps_and_clipping(N):
    rng = default_rng(21)
    shape = (180, 260)
    polygons, values = [], []
    for _ in range(300):
        cx, cy = int(rng.integers(-10, 270)), int(rng.integers(-10, 190))
        n = int(rng.integers(3, 9))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        rad = rng.uniform(3, 40, n)
        polygons.append(np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], axis=1).round().astype(np.int32))
        values.append(float(rng.uniform(1, 50)))
    mask = np.zeros(shape, np.uint8)
    score = np.zeros(shape, np.float32)
    N.paint_polys(polygons, values=values, mask=mask, score=score)
    want_mask, want_score = _paint_reference(shape, polygons, values)
    np.testing.assert_array_equal(mask, want_mask)
    np.testing.assert_array_equal(score, want_score)
    # mask only, onto a plane that already has content: untouched outside the polygons
    pre = (rng.random(shape) < 0.1).astype(np.uint8)
    out = pre.copy()
    N.paint_polys(polygons[:5], mask=out)
    np.testing.assert_array_equal(out, pre | _paint_reference(shape, polygons[:5], [1] * 5)[0])
    # nothing to paint is a no-op
    N.paint_polys([], mask=out)


def test_paint_polys_page_of_char_boxes(N):
    """Thousands of small quadrilaterals (char boxes) on a 2048^2 page, heights painted large to small."""
    rng = default_rng(22)
    shape = (2048, 2048)
    polygons, values = [], []
    for row in range(60):
        y = 20 + row * 33
        x = 15
        while x < 2000:
            w, h = int(rng.integers(8, 30)), int(rng.integers(14, 30))
            skew = int(rng.integers(-3, 4))
            polygons.append(np.asarray([(x + skew, y), (x + w + skew, y + 1), (x + w, y + h), (x, y + h - 1)], np.int32))
            values.append(float(h) + float(rng.random()))
            x += w - int(rng.integers(0, 4))  # neighbours overlap a little
    order = np.argsort(values)[::-1]
    polygons = [polygons[i] for i in order]
    values = [values[i] for i in order]
    assert len(polygons) > 3000
    mask = np.zeros(shape, np.uint8)
    score = np.zeros(shape, np.float32)
    N.paint_polys(polygons, values=values, mask=mask, score=score)
    want_mask, want_score = _paint_reference(shape, polygons, values)
    np.testing.assert_array_equal(mask, want_mask)
    np.testing.assert_array_equal(score, want_score)


# ------------------------------------------------------------------ pipeline steps
from vkit_amd.pipeline.text_detection.synthetic_page import synthetic_page_input as _synthetic_page_input  # noqa: E402


def _assembled_reference(step_input):
    """The page of ``PageAssemblerStep.run`` for ``step_input``, layer by layer with the oracle, in the reference's order
    (pipeline/text_detection/page_assembler.py:155-236)."""
    from vkit_amd.mechanism.distortion import rotate
    # the same layer list applied one by one with the oracle, in the reference's order
    want = step_input.page_background_step_output.background_image.mat.copy()
    full = (0, 0) + want.shape[:2]
    for page_image in step_input.page_image_step_output.page_image_collection.page_images:
        b = page_image.box
        O.fill(want, (b.up, b.left, b.height, b.width), page_image.image.mat, alpha=page_image.alpha)
    for score_map in step_input.page_barcode_step_output.barcode_qr_score_maps:
        O.fill(want, full, (0, 0, 0), alpha=score_map.mat)
    out = step_input.page_text_line_bounding_box_step_output
    for score_map, color in zip(out.score_maps, out.colors):
        O.fill(want, full, color, alpha=score_map.mat)
    for text_line in step_input.page_text_line_step_output.page_text_line_collection.text_lines:
        b = text_line.box
        geo = (b.up, b.left, b.height, b.width)
        if text_line.score_map:
            O.fill(want, geo, text_line.glyph_color, alpha=text_line.score_map.mat)
        else:
            O.fill(want, geo, text_line.image.mat, mask=text_line.mask.mat)
    symbols = step_input.page_non_text_symbol_step_output
    for image, b, alpha in zip(symbols.images, symbols.boxes, symbols.alphas):
        O.fill(want, (b.up, b.left, b.height, b.width), image.mat, alpha=alpha)
    seals = step_input.page_text_line_step_output.page_seal_impression_text_line_collection
    for seal, resource in zip(seals.seal_impressions, seals.seal_impression_resources):
        rotated = rotate.distort({'angle': resource.angle}, mask=seal.background_mask,
                                 score_map=resource.text_line_filled_score_map)
        center = resource.box.get_center_point()
        up = center.y - rotated.mask.height // 2
        left = center.x - rotated.mask.width // 2
        geo = (up, left, rotated.mask.height, rotated.mask.width)
        O.fill(want, geo, seal.color, mask=rotated.mask.mat, alpha=seal.alpha)
        O.fill(want, geo, seal.color, alpha=rotated.score_map.mat)
    return want


@pytest.mark.parametrize('size,n_lines', [(256, 24), (1024, 64)])      # (1024, 64): BASELINE config 3's page, 64 text layers
def test_page_assembler_layer_order(N, size, n_lines):
    from vkit_amd.pipeline import text_detection as T
    step_input = _synthetic_page_input(seed=7, size=size, n_lines=n_lines)
    page = T.page_assembler_step_factory.create().run(step_input, default_rng(0)).page
    want = _assembled_reference(step_input)
    np.testing.assert_array_equal(page.image.mat, want)
    assert len(page.page_seal_impression_char_polygon_collection.char_polygons) == 1
    assert not page.image.mat.flags.writeable
    assert (page.image.mat != step_input.page_background_step_output.background_image.mat).any()


def test_page_assembler_many_pages_across_ring_wraps(N):
    """Thirty C4 pages assembled back to back without a read in between: every page stages ~8 MB of layer planes in the context's
    page-locked ring (one copy on the copy stream per page), so the ring wraps -- and may grow -- several times while earlier
    composites are still queued; every page must come out as the oracle's layer-by-layer result."""
    from vkit_amd.pipeline import text_detection as T
    assembler = T.page_assembler_step_factory.create()
    inputs = [_synthetic_page_input(seed=100 + k, size=1024, n_lines=64) for k in range(6)]
    wants = [_assembled_reference(si) for si in inputs]
    pages = [assembler.run(inputs[k % 6], default_rng(k)).page for k in range(30)]
    for k, page in enumerate(pages):
        np.testing.assert_array_equal(page.image.mat, wants[k % 6], err_msg=f'page {k}')


@pytest.mark.parametrize('size,n_lines,seeds', [(256, 24, 6), (1024, 64, 2)])   # (1024, 64): BASELINE config 3's page
def test_page_distortion_step(N, size, n_lines, seeds):
    from oracle_replay import REPLAYABLE, replay
    from vkit_amd.mechanism.distortion_policy import random_distortion_factory
    from vkit_amd.mechanism.distortion_policy.random_distortion import RandomDistortionDebug
    from vkit_amd.element import Mask, PointList
    from vkit_amd.pipeline import text_detection as T
    step_input = _synthetic_page_input(seed=9, size=size, n_lines=n_lines)
    page_output = T.page_assembler_step_factory.create().run(step_input, default_rng(0))
    # every policy the oracle can replay from (config, state) stays enabled -- the two pass-through members included;
    # the members whose random planes are rebuilt elsewhere (test_gpu_pointwise.py) are switched off here
    all_names = [p.name for st in random_distortion_factory.create().stages for p in st.config.distortion_policies]
    factory_config = {'disabled_policy_names': sorted(set(all_names) - set(REPLAYABLE)), 'prob_geometric': 1.0,
                      'num_photometric_min': 1}
    step = T.page_distortion_step_factory.create({'random_distortion_factory_config': factory_config})
    shapes = set()
    replayed = set()
    for seed in range(seeds):
        out = step.run(T.PageDistortionStepInput(page_output), default_rng(seed))
        shapes.add(out.page_image.shape)

        # the same chain, driven directly
        page = page_output.page
        active = np.ones(page.image.shape, np.uint8)
        active[0] = active[-1] = 0
        active[:, 0] = active[:, -1] = 0
        chars = page.page_char_polygon_collection
        lines = page.page_text_line_polygon_collection
        flat_polygons = (tuple(chars.char_polygons) + tuple(chars.adjusted_char_polygons) + tuple(lines.polygons)
                         + tuple(page.page_disconnected_text_region_collection.to_polygons())
                         + tuple(page.page_non_text_region_collection.to_polygons())
                         + tuple(page.page_seal_impression_char_polygon_collection.char_polygons))
        flat_points = PointList(tuple(chars.height_points_up) + tuple(chars.height_points_down)
                                + tuple(lines.height_points_up) + tuple(lines.height_points_down))
        result = random_distortion_factory.create(factory_config).distort(
            image=page.image, mask=Mask(mat=active), polygons=flat_polygons, points=flat_points,
            rng=default_rng(seed))
        np.testing.assert_array_equal(out.page_active_mask.mat, result.mask.mat)
        # ... and the chain itself against the oracle: every distortion replayed from the config / state it reports
        debug = RandomDistortionDebug()
        again = random_distortion_factory.create(factory_config).distort(
            image=page.image, mask=Mask(mat=active), polygons=flat_polygons, points=flat_points,
            rng=default_rng(seed), debug=debug)
        np.testing.assert_array_equal(again.image.mat, result.image.mat)
        cur_image, cur_mask = page.image.mat, active
        for k, name in enumerate(debug.distortion_names):
            cur_image, cur_mask, _ = replay(name, debug.distortion_configs[k], debug.distortion_states[k], cur_image,
                                            cur_mask)
            np.testing.assert_array_equal(debug.distortion_images[k].mat, cur_image, err_msg=f'{seed} {k} {name}')
            replayed.add(name)
        # (the step injects no corner points, so nothing is trimmed after the last stage)
        np.testing.assert_array_equal(result.image.mat, cur_image)
        np.testing.assert_array_equal(result.mask.mat, cur_mask)
        want = cur_image.copy()
        bottom = page.page_bottom_layer_image.mat
        if bottom.shape != want.shape:
            bottom = O.resize_cubic(bottom, want.shape[:2])
        O.fill(want, (0, 0) + want.shape[:2], bottom, mask=(result.mask.mat == 0).astype(np.uint8))
        np.testing.assert_array_equal(out.page_image.mat, want)

        # labels: sequential fills, polygon by polygon, with the oracle raster
        n_chars, n_lines = len(chars.char_polygons), len(lines.polygons)
        res_polys = result.polygons
        char_polys = res_polys[:n_chars]
        line_polys = res_polys[2 * n_chars:2 * n_chars + n_lines]
        res_points = result.points
        n_cp = len(chars.height_points_up)
        n_lp = len(lines.height_points_up)
        cu = PointList(res_points[:n_cp]).to_smooth_np_array()
        cd = PointList(res_points[n_cp:2 * n_cp]).to_smooth_np_array()
        lu = PointList(res_points[2 * n_cp:2 * n_cp + n_lp]).to_smooth_np_array()
        ld = PointList(res_points[2 * n_cp + n_lp:]).to_smooth_np_array()
        char_h = np.linalg.norm(cd - cu, axis=1) + 1
        line_h_pts = np.linalg.norm(ld - lu, axis=1) + 1
        line_h, begin = [], 0
        for g in lines.height_points_group_sizes:
            line_h.append(float(line_h_pts[begin:begin + g].mean()))
            begin += g

        def pts_of(polygon):
            b = polygon.bounding_box
            return polygon.self_relative_polygon.to_np_array() + np.asarray([b.left, b.up], np.int32)

        shape = want.shape[:2]
        m, s = _paint_reference(shape, [pts_of(p) for p in line_polys], line_h)
        np.testing.assert_array_equal(out.page_text_line_mask.mat, m)
        np.testing.assert_array_equal(out.page_text_line_height_score_map.mat, s)
        assert out.page_text_line_heights == line_h
        order = tuple(reversed(char_h.argsort()))
        m, s = _paint_reference(shape, [pts_of(char_polys[i]) for i in order], [float(char_h[i]) for i in order])
        np.testing.assert_array_equal(out.page_char_mask.mat, m)
        np.testing.assert_array_equal(out.page_char_height_score_map.mat, s)
        assert out.page_char_heights == [float(v) for v in char_h]
        assert out.page_seal_impression_char_mask.mat.sum() > 0
    assert len(shapes) > 1  # geometric distortions did change the page shape, so the resize branch ran
    assert len(replayed) >= (6 if seeds >= 6 else 2), replayed


@pytest.mark.parametrize('src_shape,dst_shape', [((20, 30), (33, 47)), ((64, 64), (32, 32)), ((97, 131), (40, 55)),
                                                 ((5, 7), (50, 3)), ((1, 1), (9, 9)), ((300, 200), (113, 77)),
                                                 ((40, 60), (40, 60))])
def test_resize_linear_and_nearest(N, src_shape, dst_shape):
    rng = default_rng(13)
    for cn in (1, 3, 4):
        shape = src_shape if cn == 1 else src_shape + (cn,)
        src = rng.integers(0, 256, shape, dtype=np.uint8)
        np.testing.assert_array_equal(N.resize(src, dst_shape, N.INTER_LINEAR), O.resize_linear(src, dst_shape))
        np.testing.assert_array_equal(N.resize(src, dst_shape, N.INTER_NEAREST), O.resize_nearest(src, dst_shape))
        np.testing.assert_array_equal(N.resize(src, dst_shape, N.INTER_CUBIC), O.resize_cubic(src, dst_shape))


def test_pixelation_operator(N):
    from vkit_amd.element import Image
    from vkit_amd.mechanism import distortion as D
    rng = default_rng(14)
    image = Image(mat=rng.integers(0, 256, (257, 311, 3), dtype=np.uint8))
    for ratio in (0.31, 0.5, 0.77, 0.999):
        np.testing.assert_array_equal(D.pixelation.distort({'ratio': ratio}, image=image).image.mat,
                                      O.pixelation(image.mat, ratio))
    big = Image(mat=rng.integers(0, 256, (2048, 2048, 3), dtype=np.uint8))
    np.testing.assert_array_equal(D.pixelation.distort({'ratio': 0.5}, image=big).image.mat, O.pixelation(big.mat, 0.5))
    np.testing.assert_array_equal(image.to_resized_image(resized_height=100, cv_resize_interpolation=1).mat,
                                  O.resize_linear(image.mat, (100, round(100 * 311 / 257))))


def test_fill_batch_matches_per_page_calls(N):
    """vkx_fill_u8_batch_dev: the layer lists of several equally shaped device pages in one launch equal the per-page calls
    (and the oracle's sequential fills) -- pages with many layers, one layer, no layer; shared alpha planes."""
    import ctypes
    ctx, lib = N.default_ctx(), N.lib()
    rng = default_rng(55)
    h, w, cn = 150, 210, 3
    n_pages = 6
    pages = [rng.integers(0, 256, (h, w, cn), dtype=np.uint8) for _ in range(n_pages)]
    counts = [9, 1, 0, 17, 2, 30]
    keep, specs, layer_begin = [], [], [0]
    total = sum(counts)
    layers = (N.VkxLayer * max(total, 1))()
    k = 0
    for p, cnt in enumerate(counts):
        for _ in range(cnt):
            bh, bw = int(rng.integers(1, h)), int(rng.integers(1, w))
            up, left = int(rng.integers(0, h - bh + 1)), int(rng.integers(0, w - bw + 1))
            kind = int(rng.integers(3))
            L = layers[k]
            L.up, L.left, L.height, L.width = up, left, bh, bw
            color = tuple(int(v) for v in rng.integers(0, 256, 3))
            for c in range(3):
                L.value_const[c] = color[c]
            alpha = mask = None
            if kind == 0:
                alpha = (rng.random((bh, bw), dtype=np.float32) * (rng.random((bh, bw)) < 0.5)).astype(np.float32)
                d = ctx.malloc(alpha.nbytes); ctx.upload(d, alpha); keep.append(d)
                L.alpha, L.alpha_stride_el, L.alpha_scalar = d, bw, 1.0
            elif kind == 1:
                mask = (rng.random((bh, bw)) < 0.4).astype(np.uint8)
                d = ctx.malloc(mask.nbytes); ctx.upload(d, mask); keep.append(d)
                L.mask, L.mask_stride, L.alpha_scalar = d, bw, float(rng.choice([1.0, 0.35]))
            else:
                L.alpha_scalar = float(rng.choice([1.0, 0.6, 0.0]))
            specs.append((p, (up, left, bh, bw), color, alpha, mask, float(L.alpha_scalar)))
            k += 1
        layer_begin.append(k)
    d_batch = [ctx.malloc(pg.nbytes) for pg in pages]
    d_single = [ctx.malloc(pg.nbytes) for pg in pages]
    for dp, ds, pg in zip(d_batch, d_single, pages):
        ctx.upload(dp, pg); ctx.upload(ds, pg)
    ptrs = (ctypes.c_void_p * n_pages)(*d_batch)
    begin = np.asarray(layer_begin, np.int32)
    N.check(lib.vkx_fill_u8_batch_dev(ctx.handle, ptrs, n_pages, h, w, cn, w * cn, layers, begin.ctypes.data))
    for p in range(n_pages):
        if counts[p]:
            sub = ctypes.cast(ctypes.byref(layers[layer_begin[p]]), ctypes.POINTER(N.VkxLayer))
            N.check(lib.vkx_fill_u8_dev(ctx.handle, d_single[p], h, w, cn, w * cn, sub, counts[p]))
    for p in range(n_pages):
        got_b, got_s = np.empty_like(pages[p]), np.empty_like(pages[p])
        ctx.download(d_batch[p], got_b); ctx.download(d_single[p], got_s)
        ctx.sync()
        np.testing.assert_array_equal(got_b, got_s, err_msg=f'page {p}')
        want = pages[p].copy()
        for (pp, box, color, alpha, mask, a) in specs:
            if pp == p:
                O.fill(want, box, color, mask=mask, alpha=alpha if alpha is not None else a)
        np.testing.assert_array_equal(got_b, want, err_msg=f'page {p} vs oracle')
    for d in keep + d_batch + d_single:
        ctx.free(d)


def test_chain_batch_assembles_pages_on_the_device(N):
    """ChainBatch.set_layers: the layer lists of all pages are composited onto the device-resident sources by one batched
    launch at every run, then the chain runs -- sources and results equal the oracle's sequential fills + chain."""
    from vkit_amd.batch import ChainBatch
    from vkit_amd.mechanism import distortion as D
    from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam
    rng = default_rng(77)
    h, w = 200, 260
    gen = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), 5)
    batch = ChainBatch()
    specs = []
    for p in range(5):
        state = D.camera_cubic_curve.generate_state(gen((h, w), default_rng(p)), (h, w))
        noise = rng.integers(-30, 30, tuple(state.result_shape) + (3,)).astype(np.int16)
        batch.add(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), state, blur_sigma=1.0, hue_delta=20 * p - 37, noise=noise)
        layers, plan = [N.make_layer((0, 0, h, w), 3, (200 - p, 190, 180))], [((0, 0, h, w), (200 - p, 190, 180), None, 1.0)]
        for _ in range(int(rng.integers(0, 12))):
            bh, bw = int(rng.integers(1, h)), int(rng.integers(1, w))
            box = (int(rng.integers(0, h - bh + 1)), int(rng.integers(0, w - bw + 1)), bh, bw)
            kind = int(rng.integers(3))
            if kind == 0:
                alpha = (rng.random((bh, bw), dtype=np.float32) * (rng.random((bh, bw)) < 0.5)).astype(np.float32)
                color = tuple(int(v) for v in rng.integers(0, 256, 3))
                layers.append(N.make_layer(box, 3, color, alpha=alpha)); plan.append((box, color, None, alpha))
            elif kind == 1:
                mask = (rng.random((bh, bw)) < 0.4).astype(np.uint8)
                value = rng.integers(0, 256, (bh, bw, 3), dtype=np.uint8)
                layers.append(N.make_layer(box, 3, value, mask=mask, alpha=0.7)); plan.append((box, value, mask, 0.7))
            else:
                color = tuple(int(v) for v in rng.integers(0, 256, 3))
                layers.append(N.make_layer(box, 3, color, alpha=0.45)); plan.append((box, color, None, 0.45))
        if p != 3:                      # page 3 keeps its uploaded source: no layers
            batch.set_layers(p, layers)
        specs.append((state, noise, plan if p != 3 else None))
    uploaded3 = batch.source(3).copy()
    for rep in range(2):                # the second run starts from the composited pages: the background layer re-initialises
        batch.run()
        for p, (state, noise, plan) in enumerate(specs):
            if plan is None:
                want_src = uploaded3
            else:
                want_src = np.zeros((h, w, 3), np.uint8)
                for box, value, mask, alpha in plan:
                    O.fill(want_src, box, value, mask=mask, alpha=alpha)
            np.testing.assert_array_equal(batch.source(p), want_src, err_msg=f'source {p} run {rep}')
            mx, my = O.grid_to_map(state.src_image_grid.vertices, state.dst_image_grid.vertices, state.result_shape)
            want = O.add_noise_i16(O.color_shift_rgb(O.gaussian_blur(O.remap(want_src, mx, my), 5, 1.0), 20 * p - 37), noise)
            np.testing.assert_array_equal(batch.result(p), want, err_msg=f'result {p} run {rep}')
    batch.close()
