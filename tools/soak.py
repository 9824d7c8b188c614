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
counts = dict(chain=0, resize=0, poly=0, fill=0, remap=0)
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
        counts['resize'] += 3
    # polygons
    for _ in range(40):
        h, w = int(rng.integers(1, 150)), int(rng.integers(1, 150))
        n = int(rng.integers(1, 14))
        pts = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], axis=1)
        assert (N.fill_poly_mask((h, w), pts) == O.fill_poly((h, w), pts.astype(np.int32))).all()
        counts['poly'] += 1
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
