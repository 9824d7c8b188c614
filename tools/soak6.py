#!/usr/bin/env python3
"""Sixth soak: the round-2 additions against the oracle on random inputs -- cv.ellipse outlines (any centre, axes, thickness,
clipped or outside), similarity_mls lattice projection with 2 .. 1500 handles, throughput-mode noise planes, and the
overlapped host pipeline against the synchronous calls.  Usage: tools/soak6.py <seconds> <seed>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from numpy.random import default_rng

import oracle as O
from vkit_amd import _native as N
from vkit_amd.hostpipe import HostPipeline
from vkit_amd.mechanism import distortion as D
from vkit_amd.mechanism.distortion_policy.geometric import mls as P_mls

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
counts = {'ellipse': 0, 'mls': 0, 'noise': 0, 'pipeline': 0}
pipe = HostPipeline(depth=4, lanes=4)
while time.time() - t0 < budget:
    kind = int(rng.integers(4))
    if kind == 0:
        h, w = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        center = (int(rng.integers(-50, w + 50)), int(rng.integers(-50, h + 50)))
        n = int(rng.integers(1, 12))
        axes = [(int(rng.integers(0, 2 * w + 2)), int(rng.integers(0, 2 * h + 2))) for _ in range(n)]
        if rng.random() < 0.3:
            axes = [(int(a * 0.05), int(b * 0.05)) for a, b in axes]            # the coarse arc steps
        thickness = int(rng.integers(1, 9))
        got = (rng.random((h, w)) < 0.02).astype(np.uint8) * 7
        want = got.copy()
        N.ellipse_mask(got, center, axes, thickness)
        for a in axes:
            O.ellipse_outline(want, center, a, thickness)
        assert (got == want).all(), ('ellipse', (h, w), center, axes, thickness)
        counts['ellipse'] += 1
    elif kind == 1:
        n = int(rng.choice([2, 3, 4, 5, 8, 25, 48, 49, 64, 128, 129, 300, int(rng.integers(2, 1500))]))
        span = int(rng.integers(50, 4000))
        ps = rng.integers(0, span, (n, 2)).astype(np.float64) + 0.25 * (np.arange(n) % 3)[:, None]
        qs = ps + rng.normal(0, span * 0.01 + 1, (n, 2))
        p, q = np.rint(ps).astype(np.float32), np.rint(qs).astype(np.float32)
        V = rng.integers(0, span, (int(rng.integers(1, 3000)), 2)).astype(np.float64) + 0.5
        assert (N.mls_project(p, q, ps, qs, V) == O.mls_project(p, q, ps, qs, V)).all(), ('mls', n, span)
        counts['mls'] += 1
    elif kind == 2:
        h, w, cn = int(rng.integers(1, 300)), int(rng.integers(1, 300)), int(rng.choice([1, 3, 4]))
        std = float(rng.choice([0.3, 1.0, 5.0, 10.0, 27.0, 28.0, 40.0, 300.0]))
        seed = int(rng.integers(1 << 62))
        shape = (h, w) if cn == 1 else (h, w, cn)
        assert (N.noise_normal_i16(shape, std, seed) == O.noise_normal_i16(shape, std, seed)).all(), ('noise', shape, std, seed)
        counts['noise'] += 1
    else:
        size = (int(rng.integers(40, 300)), int(rng.integers(40, 300)))
        gen = P_mls.SimilarityMlsConfigGenerator(P_mls.SimilarityMlsConfigGeneratorConfig(), int(rng.integers(1, 11)))
        state = D.similarity_mls.generate_state(gen(size, default_rng(int(rng.integers(1 << 30)))), size)
        image = rng.integers(0, 256, size + (3,), dtype=np.uint8)
        mask = (rng.random(size) < 0.5).astype(np.uint8)
        score = rng.random(size, dtype=np.float32)
        t1 = pipe.submit_remap([image, mask, score], state)
        noise = rng.integers(-40, 40, tuple(state.result_shape) + (3,)).astype(np.int16)
        sigma, delta = float(rng.choice([0.7, 1.0, 2.0])), int(rng.integers(-255, 256))
        t2 = pipe.submit_chain(image, state, blur_sigma=sigma, hue_delta=delta, noise=noise)
        mx, my = O.grid_to_map(state.src_image_grid.vertices, state.dst_image_grid.vertices, state.result_shape)
        got = pipe.result(t1)
        for g, src in zip(got, (image, mask, score)):
            want = O.remap(src, mx, my)
            assert (g.view(np.uint32) == want.view(np.uint32)).all() if g.dtype == np.float32 else (g == want).all(), 'pipeline remap'
        k = max(3, round(3 * sigma) + 1)
        k += 1 - k % 2
        want = O.add_noise_i16(O.color_shift_rgb(O.gaussian_blur(O.remap(image, mx, my), k, sigma), delta), noise)
        assert (pipe.result(t2)[0] == want).all(), 'pipeline chain'
        counts['pipeline'] += 1
pipe.close()
print('soak6 ok', counts, round(time.time() - t0), 's')
