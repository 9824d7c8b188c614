#!/usr/bin/env python3
"""One-off soak on the GPU box: the randomised parity checks of the test-suite with many more seeds."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from numpy.random import default_rng

import oracle as O
import test_gpu_parity as T
from vkit_amd import _native as N

t0 = time.time()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
counts = dict(chain=0, resize=0, poly=0, fill=0, remap=0, filter2d=0, points=0)
while time.time() - t0 < budget:
    # fused chain, ragged batch
    grids, names = {}, []
    for i in range(8):
        h, w = int(rng.integers(2, 400)), int(rng.integers(2, 400))
        sv, dv, ds = T.synthetic_grid(h, w, int(rng.integers(4, 50)), float(rng.uniform(0, 16)), seed=int(rng.integers(1 << 30)))
        if rng.random() < 0.3:
            dv = dv + np.asarray([int(rng.integers(0, 120)), int(rng.integers(0, 120))], np.int32)
            ds = (int(dv[..., 1].max()) + 1 + int(rng.integers(0, 100)), int(dv[..., 0].max()) + 1 + int(rng.integers(0, 100)))
        grids[f'g{i}'] = (sv, dv, ds, (h, w))
        names.append(f'g{i}')
    sig = [None, 0.5, 0.7, 1.0, 1.4, 2.0]
    T._chain_case(N, grids, names, int(rng.integers(1 << 30)), [sig[int(k)] for k in rng.integers(0, 6, 8)],
                  [None if k % 4 == 0 else int(k) - 128 for k in rng.integers(0, 256, 8)], [bool(k) for k in rng.integers(0, 2, 8)])
    counts['chain'] += 8
    # multi-element remap through one lattice
    sv, dv, ds, (h, w) = grids['g0']
    mats = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8), rng.integers(0, 2, (h, w), dtype=np.uint8),
            rng.random((h, w), dtype=np.float32), rng.integers(0, 256, (h, w, 4), dtype=np.uint8)]
    mx, my = O.grid_to_map(sv, dv, ds)
    for got, m in zip(N.grid_remap(mats, sv, dv, ds), mats):
        assert (got == O.remap(m, mx, my)).all()
    counts['remap'] += 4
    # resizes
    for _ in range(10):
        sh, sw, dh, dw = (int(v) for v in rng.integers(1, 300, 4))
        cn = int(rng.choice([1, 3, 4]))
        src = rng.integers(0, 256, (sh, sw) if cn == 1 else (sh, sw, cn), dtype=np.uint8)
        assert (N.resize(src, (dh, dw), 2) == O.resize_cubic(src, (dh, dw))).all()
        assert (N.resize(src, (dh, dw), 1) == O.resize_linear(src, (dh, dw))).all()
        assert (N.resize(src, (dh, dw), 0) == O.resize_nearest(src, (dh, dw))).all()
        for inter in (4, 5, 6):
            assert (N.resize(src, (dh, dw), inter) == O.resize(src, (dh, dw), inter)).all()
        if dh <= sh and dw <= sw:
            assert (N.resize(src, (dh, dw), 3) == O.resize(src, (dh, dw), 3)).all()
        if cn == 1:
            plane = rng.random((sh, sw), dtype=np.float32)
            for inter in (2, 4, 5, 6):
                got, want = N.resize(plane, (dh, dw), inter), O.resize(plane, (dh, dw), inter)
                assert ((got == want) | ((got != got) & (want != want))).all()      # NaN taps: see the regression test
        counts['resize'] += 7
    # polygons
    for _ in range(40):
        h, w = int(rng.integers(1, 150)), int(rng.integers(1, 150))
        n = int(rng.integers(1, 14))
        pts = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], axis=1)
        assert (N.fill_poly_mask((h, w), pts) == O.fill_poly((h, w), pts.astype(np.int32))).all()
        counts['poly'] += 1
    # filter2D with random kernels
    for _ in range(6):
        h, w = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        cn = int(rng.choice([1, 3, 4]))
        img = rng.integers(0, 256, (h, w) if cn == 1 else (h, w, cn), dtype=np.uint8)
        kh, kw = int(rng.integers(1, 16)), int(rng.integers(1, 16))
        kernel = (rng.random((kh, kw)) * (rng.random((kh, kw)) < 0.7)).astype(np.float32)
        kernel /= max(float(kernel.sum()), 1e-3)
        assert (N.filter2d(img, kernel) == O.filter2d(img, kernel)).all()
        counts['filter2d'] += 1
    # point projection through the first lattice
    sv, dv, ds, (h, w) = grids['g0']
    gs = int(sv[1, 0, 1] - sv[0, 0, 1]) if sv.shape[0] > 2 else max(h, w)
    if sv.shape[0] > 2 and sv.shape[1] > 2:
        n = 300
        pts = np.stack([rng.integers(0, (sv.shape[1] - 1) * gs, n), rng.integers(0, (sv.shape[0] - 1) * gs, n)], axis=1)
        smooth = pts + rng.uniform(-0.5, 0.5, pts.shape)
        got = N.project_points(sv, dv, gs, pts, smooth)
        want = O.grid_project_points(sv, dv, gs, pts, smooth)
        assert (np.abs(got - want) <= 4 * np.spacing(np.abs(want)) + 1e-300).all()
        counts['points'] += n
    # composite list
    page = rng.integers(0, 256, (150, 200, 3), dtype=np.uint8)
    want = page.copy()
    layers = []
    for _ in range(25):
        bh, bw = int(rng.integers(1, 80)), int(rng.integers(1, 120))
        up, left = int(rng.integers(0, 150 - bh + 1)), int(rng.integers(0, 200 - bw + 1))
        kind = int(rng.integers(0, 3))
        value = tuple(int(v) for v in rng.integers(0, 256, 3)) if rng.random() < 0.5 else rng.integers(0, 256, (bh, bw, 3), dtype=np.uint8)
        mask = (rng.random((bh, bw)) < 0.5).astype(np.uint8) if kind == 0 else None
        alpha = (rng.random((bh, bw), dtype=np.float32) * (rng.random((bh, bw)) < 0.5)).astype(np.float32) if kind == 1 else float(rng.choice([1.0, rng.random()]))
        layers.append(N.make_layer((up, left, bh, bw), 3, value, mask=mask, alpha=alpha))
        O.fill(want, (up, left, bh, bw), value, mask=mask, alpha=alpha)
    N.fill(page, layers)
    assert (page == want).all()
    counts['fill'] += 25
print('soak ok', counts, round(time.time() - t0), 's')
