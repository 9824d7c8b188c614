#!/usr/bin/env python3
"""Third soak: ordered polygon painting (vkx_paint_polys) with random, partly self-intersecting and partly off-plane
polygons against the sequential oracle fills.  Usage: tools/soak3.py <seconds> <seed>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from numpy.random import default_rng

from vkit_amd import _native as N
from test_gpu_composite import _paint_reference

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
n_polys = n_pages = 0
while time.time() - t0 < budget:
    h, w = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    polygons, values = [], []
    for _ in range(int(rng.integers(0, 120))):
        n = int(rng.integers(1, 12))
        kind = rng.random()
        if kind < 0.5:          # star-shaped around a centre that may lie off the plane
            cx, cy = int(rng.integers(-20, w + 20)), int(rng.integers(-20, h + 20))
            ang = np.sort(rng.uniform(0, 2 * np.pi, n))
            rad = rng.uniform(0, 60, n)
            pts = np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], axis=1).round().astype(np.int32)
        elif kind < 0.8:        # arbitrary vertex order: self intersections, even-odd holes
            pts = np.stack([rng.integers(-15, w + 15, n), rng.integers(-15, h + 15, n)], axis=1).astype(np.int32)
        else:                   # thin / degenerate: repeated vertices, horizontal and vertical runs
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            pts = np.array([[x0, y0], [x0 + int(rng.integers(0, 40)), y0], [x0 + int(rng.integers(0, 40)), y0],
                            [x0, y0 + int(rng.integers(0, 3))]], np.int32)
        polygons.append(pts)
        values.append(float(rng.uniform(1, 50)))
    mask = np.zeros((h, w), np.uint8)
    score = np.zeros((h, w), np.float32)
    N.paint_polys(polygons, values=values, mask=mask, score=score)
    want_mask, want_score = _paint_reference((h, w), polygons, values)
    assert (mask == want_mask).all() and (score == want_score).all(), (h, w, len(polygons))
    n_polys += len(polygons)
    n_pages += 1
# composite layer lists on every destination type: uint8 x 1 / 3 / 4 channels and float32, plain / keep-max / keep-min,
# scalar and plane alpha, masks, constant and plane values
import oracle as O
t1 = time.time()
n_layers = 0
while time.time() - t1 < budget / 2:
    h, w = int(rng.integers(1, 200)), int(rng.integers(1, 260))
    kind = int(rng.integers(0, 4))
    if kind == 3:
        page = (rng.random((h, w), dtype=np.float32) * 30).astype(np.float32)
        cn, dtype = 1, np.float32
    else:
        cn = (1, 3, 4)[kind]
        page = rng.integers(0, 256, (h, w) if cn == 1 else (h, w, cn), dtype=np.uint8)
        dtype = np.uint8
    want = page.copy()
    layers = []
    for _ in range(int(rng.integers(0, 40))):
        bh, bw = int(rng.integers(1, h + 1)), int(rng.integers(1, w + 1))
        up, left = int(rng.integers(0, h - bh + 1)), int(rng.integers(0, w - bw + 1))
        mode = int(rng.choice([0, 0, 1, 2]))
        shape = (bh, bw) if cn == 1 else (bh, bw, cn)
        if rng.random() < 0.5:
            value = (rng.random(shape, dtype=np.float32) * 40).astype(np.float32) if dtype == np.float32 \
                else rng.integers(0, 256, shape, dtype=np.uint8)
        elif dtype == np.float32:
            value = float(rng.uniform(0, 40))
        else:
            value = tuple(int(v) for v in rng.integers(0, 256, cn)) if cn > 1 else int(rng.integers(0, 256))
        mask = (rng.random((bh, bw)) < 0.6).astype(np.uint8) if rng.random() < 0.4 else None
        if mask is None and rng.random() < 0.4:
            alpha = (rng.random((bh, bw), dtype=np.float32) * (rng.random((bh, bw)) < 0.6)).astype(np.float32)
        elif mask is not None and cn == 1 and dtype == np.uint8:
            alpha = 1.0           # 2-D uint8 + mask + fractional scalar alpha raises in the reference
        else:
            alpha = float(rng.choice([1.0, 1.0, rng.random()]))
        layers.append(N.make_layer((up, left, bh, bw), cn, value, mask=mask, alpha=alpha, mode=mode, dtype=dtype))
        O.fill(want, (up, left, bh, bw), value, mask=mask, alpha=alpha, mode=mode)
    N.fill(page, layers)
    assert (page == want).all(), (h, w, cn, dtype, len(layers))
    n_layers += len(layers)
print('soak3 ok', n_pages, 'planes', n_polys, 'polygons', n_layers, 'composite layers', round(time.time() - t0), 's')
