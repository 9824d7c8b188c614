"""Round 6's dispatch diet of the page path (profiles/r6a_page_dispatches.txt -> r6d): the variants it introduced against the forms they
replace and against the oracle -- fresh-plane paint, layer planes staged through the page-locked ring (cropped score maps, planes already
on the device, planes read in place), the lookup table read from the ring, point projection on a mapped block."""
import os
import subprocess
import sys

import numpy as np
import pytest
from numpy.random import default_rng

import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def N():
    from vkit_amd import _native
    return _native


def _polys(rng, shape, n):
    out, values = [], []
    for _ in range(n):
        cx, cy = int(rng.integers(-10, shape[1] + 10)), int(rng.integers(-10, shape[0] + 10))
        k = int(rng.integers(3, 9))
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        rad = rng.uniform(3, 40, k)
        out.append(np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], axis=1).round().astype(np.int32))
        values.append(float(rng.uniform(1, 50)))
    return out, values


def test_fresh_paint_equals_paint_on_zeroed_planes(N):
    """vkx_paint_polys_fresh_dev on planes full of garbage == vkx_paint_polys_dev on zeroed planes; calls of different plane shapes
    alternate on one context (the ownership raster is cleared by the kernel that reads it, never by a memset), in place paints in between."""
    ctx = N.default_ctx()
    rng = default_rng(3)
    for shape in ((180, 260), (97, 131), (300, 64), (180, 260)):
        polygons, values = _polys(rng, shape, 120)
        garbage_m = ctx.to_device(rng.integers(0, 256, shape, dtype=np.uint8))
        garbage_s = ctx.to_device(rng.random(shape, dtype=np.float32))
        N.paint_polys(polygons, values=values, mask=garbage_m, score=garbage_s, fresh=True)
        zero_m, zero_s = N.dev_zeros(shape, np.uint8), N.dev_zeros(shape, np.float32)
        N.paint_polys(polygons, values=values, mask=zero_m, score=zero_s)
        np.testing.assert_array_equal(garbage_m.host(), zero_m.host())
        np.testing.assert_array_equal(garbage_s.host(), zero_s.host())
        # in place onto content: untouched outside the polygons (and the raster the fresh call left behind is clean)
        pre = (rng.random(shape) < 0.2).astype(np.uint8)
        plane = ctx.to_device(pre)
        N.paint_polys(polygons[:7], mask=plane)
        want = pre.copy()
        ref = np.zeros(shape, np.uint8)
        N.paint_polys(polygons[:7], mask=ref)
        np.testing.assert_array_equal(plane.host(), pre | ref)
        del want
    # mask only / score only, fresh
    shape = (150, 170)
    polygons, values = _polys(rng, shape, 40)
    m = ctx.to_device(rng.integers(0, 256, shape, dtype=np.uint8))
    N.paint_polys(polygons, mask=m, fresh=True)
    ref = np.zeros(shape, np.uint8)
    N.paint_polys(polygons, mask=ref)
    np.testing.assert_array_equal(m.host(), ref)
    s = ctx.to_device(rng.random(shape, dtype=np.float32))
    N.paint_polys(polygons, values=values, score=s, fresh=True)
    ref_s = np.zeros(shape, np.float32)
    N.paint_polys(polygons, values=values, score=ref_s)
    np.testing.assert_array_equal(s.host(), ref_s)


def _page_layers(N, rng, h, w):
    """A page-like layer list: a background copy, page-sized score maps that are zero but for a few rows (top / bottom / none at all),
    dense text-line alphas, mask + image layers; plan = the same for the oracle."""
    layers, plan = [], []

    def add(box, value, mask=None, alpha=1.0):
        layers.append(N.make_layer(box, 3, value, mask=mask, alpha=alpha))
        plan.append((box, value, mask, alpha))

    add((0, 0, h, w), rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    for rows in ((5, 8), (h - 9, h - 2), (0, 1), (h - 1, h), None):
        alpha = np.zeros((h, w), np.float32)
        if rows is not None:
            alpha[rows[0]:rows[1], 7:w - 11] = rng.random((rows[1] - rows[0], w - 18), dtype=np.float32)
        add((0, 0, h, w), tuple(int(v) for v in rng.integers(0, 256, 3)), alpha=alpha)
    neg = np.full((20, 30), -0.5, np.float32)          # selects nothing (alpha > 0 is the selection)
    neg[3, 4] = 0.25
    add((10, 12, 20, 30), (9, 8, 7), alpha=neg)
    for _ in range(12):
        bh, bw = int(rng.integers(4, 40)), int(rng.integers(8, w // 2))
        box = (int(rng.integers(0, h - bh)), int(rng.integers(0, w - bw)), bh, bw)
        kind = int(rng.integers(3))
        if kind == 0:
            add(box, (10, 20, 30), alpha=(rng.random((bh, bw), dtype=np.float32) * (rng.random((bh, bw)) < 0.3)).astype(np.float32))
        elif kind == 1:
            add(box, rng.integers(0, 256, (bh, bw, 3), dtype=np.uint8), mask=(rng.random((bh, bw)) < 0.4).astype(np.uint8), alpha=0.7)
        else:
            # an alpha-selected layer with an IMAGE value: the crop has to move the value rows with the alpha rows
            alpha = np.zeros((bh, bw), np.float32)
            alpha[bh // 2:, :] = rng.random((bh - bh // 2, bw), dtype=np.float32)
            add(box, rng.integers(0, 256, (bh, bw, 3), dtype=np.uint8), alpha=alpha)
    return layers, plan


@pytest.mark.parametrize('mapped', ['default', '1', '0'])
def test_host_layers_onto_a_device_page(mapped):
    """vkx_fill_u8_dev_host_layers: planes gathered in the ring and copied once (default for a page's worth), read in place
    (VKX_LAYERS_MAPPED=1), or staged as in round 5 (=0) -- all equal to the oracle's sequential fills; the env is read once per
    process, so every mode runs in a process of its own."""
    code = r'''
import sys
import numpy as np
from numpy.random import default_rng
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle as O
from vkit_amd import _native as N
from test_gpu_page_dispatch_diet import _page_layers
ctx = N.default_ctx()
rng = default_rng(17)
for h, w in ((96, 128), (201, 333), (1024, 1024)):
    layers, plan = _page_layers(N, rng, h, w)
    page = ctx.dev_empty((h, w, 3), np.uint8)            # uninitialised: the background layer covers it
    for rep in range(2):
        N.fill(page, layers)
        want = np.zeros((h, w, 3), np.uint8)
        for box, value, mask, alpha in plan:
            O.fill(want, box, value, mask=mask, alpha=alpha)
        assert (page.host() == want).all(), (h, w, rep)
print('ok')
''' % (ROOT, os.path.join(ROOT, 'tests'))
    env = dict(os.environ)
    env.pop('VKX_LAYERS_MAPPED', None)
    if mapped != 'default':
        env['VKX_LAYERS_MAPPED'] = mapped
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_layers_with_planes_already_on_the_device(N):
    """fill_page_inactive_region's shape: a device mask selects, a HOST image is the value (VKX_LAYER_MASK_ON_DEVICE); and the reverse,
    and all three planes of a layer on different sides."""
    ctx = N.default_ctx()
    rng = default_rng(29)
    h, w = 140, 200
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    mask = (rng.random((h, w)) < 0.5).astype(np.uint8)
    value = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    alpha = (rng.random((60, 80), dtype=np.float32) * (rng.random((60, 80)) < 0.6)).astype(np.float32)
    small = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    want = base.copy()
    O.fill(want, (0, 0, h, w), value, mask=mask, alpha=1.0)
    O.fill(want, (30, 40, 60, 80), small, mask=None, alpha=alpha)
    O.fill(want, (5, 6, 60, 80), (1, 2, 3), mask=mask[:60, :80].copy(), alpha=0.5)
    for dev_mask, dev_value, dev_alpha in ((True, False, False), (False, True, True), (True, True, False)):
        page = ctx.to_device(base)
        layers = [
            N.make_layer((0, 0, h, w), 3, ctx.to_device(value) if dev_value else value, mask=ctx.to_device(mask) if dev_mask else mask),
            N.make_layer((30, 40, 60, 80), 3, ctx.to_device(small) if dev_value else small, alpha=ctx.to_device(alpha) if dev_alpha else alpha),
            N.make_layer((5, 6, 60, 80), 3, (1, 2, 3), mask=ctx.to_device(mask[:60, :80].copy()) if dev_mask else mask[:60, :80].copy(), alpha=0.5),
        ]
        N.fill(page, layers)
        np.testing.assert_array_equal(page.host(), want)


def test_lookup_table_from_the_ring_and_point_projection(N):
    """apply_lut with its table read in the mapped ring (dense planes) and through device memory (strided views); a burst of calls with
    different tables on one stream (the ring keeps every table alive until its kernel has run)."""
    ctx = N.default_ctx()
    rng = default_rng(31)
    img = rng.integers(0, 256, (257, 311, 3), dtype=np.uint8)
    dev = ctx.to_device(img)
    tables = [rng.integers(0, 256, (3, 256), dtype=np.uint8) for _ in range(40)]
    outs = [N.apply_lut(dev, t) for t in tables]
    for t, o in zip(tables, outs):
        got = o.host() if hasattr(o, 'host') else o
        want = np.stack([t[c][img[..., c]] for c in range(3)], axis=-1)
        np.testing.assert_array_equal(got, want)


def test_paint_sets_equal_separate_paints(N):
    """vkx_paint_poly_sets_fresh_dev: the label plane sets of a page in one call (bands of one ownership raster) == one fresh paint per set --
    polygons reaching over the plane's top and bottom edge (they must not bleed into the neighbour band), an empty set, mask-only and
    score-only sets, one to eight sets, twice on the same context with other shapes in between."""
    ctx = N.default_ctx()
    rng = default_rng(41)
    for shape, n_sets in (((120, 200), 4), ((64, 64), 8), ((257, 131), 1), ((120, 200), 5)):
        sets, want = [], []
        for k in range(n_sets):
            polygons, values = _polys(rng, shape, 0 if k == 2 else int(rng.integers(1, 60)))
            # tall polygons crossing both horizontal edges of the plane
            for _ in range(3):
                x = int(rng.integers(0, shape[1] - 8))
                polygons.append(np.asarray([(x, -30), (x + 7, -25), (x + 9, shape[0] + 20), (x - 3, shape[0] + 33)], np.int32))
                values.append(float(rng.uniform(1, 9)))
            want_mask, want_score = (k % 3 != 1), (k % 3 != 0)
            flat = np.concatenate(polygons, axis=0).astype(np.int32)
            offsets = np.concatenate([[0], np.cumsum([len(p) for p in polygons])]).astype(np.int32)
            mask = ctx.to_device(rng.integers(0, 256, shape, dtype=np.uint8)) if want_mask else None
            score = ctx.to_device(rng.random(shape, dtype=np.float32)) if want_score else None
            sets.append((flat, offsets, values if want_score else None, mask, score))
            ref_m = ctx.dev_empty(shape, np.uint8) if want_mask else None
            ref_s = ctx.dev_empty(shape, np.float32) if want_score else None
            N.paint_polys_flat(flat, offsets, values=values if want_score else None, mask=ref_m, score=ref_s, fresh=True)
            want.append((ref_m, ref_s))
        N.paint_poly_sets_fresh(sets, shape)
        for k, ((_f, _o, _v, mask, score), (ref_m, ref_s)) in enumerate(zip(sets, want)):
            if mask is not None:
                np.testing.assert_array_equal(mask.host(), ref_m.host(), err_msg=f'mask of set {k} {shape}')
            if score is not None:
                np.testing.assert_array_equal(score.host(), ref_s.host(), err_msg=f'score of set {k} {shape}')
    # an empty polygon list: the planes come out zero
    m = ctx.to_device(rng.integers(1, 256, (40, 50), dtype=np.uint8))
    N.paint_poly_sets_fresh([(np.zeros((0, 2), np.int32), np.zeros(1, np.int32), None, m, None)], (40, 50))
    assert not m.host().any()


def test_lookup_tables_of_several_planes_in_one_launch(N):
    """vkx_apply_lut_u8_planes_dev: one to eight dense planes of different sizes (1 byte .. a megapixel, lengths that are no multiple of 16),
    each through its own table, equal to numpy's take."""
    ctx = N.default_ctx()
    rng = default_rng(53)
    for shapes in (((1, 1),), ((3, 5), (64, 64), (1, 17)), ((257, 131),) * 4, ((1024, 1024), (7, 9), (33, 1), (2, 8), (100, 101), (1, 1), (640, 3), (5, 5))):
        planes = [rng.integers(0, 256, s, dtype=np.uint8) for s in shapes]
        luts = [rng.integers(0, 256, 256, dtype=np.uint8) for _ in shapes]
        outs = N.apply_lut_planes([ctx.to_device(p) for p in planes], luts)
        for p, t, o in zip(planes, luts, outs):
            np.testing.assert_array_equal(o.host(), t[p])
    with pytest.raises(Exception):
        N.apply_lut_planes([ctx.to_device(planes[0])] * 9, [luts[0]] * 9)
