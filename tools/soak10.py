#!/usr/bin/env python3
"""Soak of k_composite_rgb (uint8 RGB pages whose rows are whole 4-pixel groups; batched launch): random page sizes, page counts, layer
kinds and stackings against the oracle's sequential fills.  VKX_RGB_RUN fixes the run length of tile slots per workgroup (read once per
process).  Usage: [VKX_RGB_RUN=k] tools/soak10.py <seconds> <seed>"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from numpy.random import default_rng

import oracle as O
from vkit_amd import _native as N

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx, lib = N.default_ctx(), N.lib()
t0 = time.time()
n_batches = n_layers = n_bad = 0
while time.time() - t0 < budget:
    h, w, cn = int(rng.integers(1, 260)), 4 * int(rng.integers(1, 90)), 3
    n_pages = int(rng.integers(1, 7))
    pages = [rng.integers(0, 256, (h, w, cn), dtype=np.uint8) for _ in range(n_pages)]
    counts = [int(rng.choice([0, 1, 3, 12, 40, 90])) for _ in range(n_pages)]
    layers = (N.VkxLayer * max(1, sum(counts)))()
    keep, specs, begin = [], [], [0]
    k = 0
    stacked = rng.random() < 0.5
    for p, cnt in enumerate(counts):
        for j in range(cnt):
            if stacked and j % 2 == 0:
                bh, bw = int(rng.integers(1, min(h, 40) + 1)), int(rng.integers(1, min(w, 200) + 1))
                up, left = int(rng.integers(0, min(24, h - bh) + 1)), int(rng.integers(0, min(50, w - bw) + 1))
            else:
                bh, bw = int(rng.integers(1, h + 1)), int(rng.integers(1, w + 1))
                up, left = int(rng.integers(0, h - bh + 1)), int(rng.integers(0, w - bw + 1))
            kind = int(rng.integers(6))
            if j == 0 and rng.random() < 0.3:
                up, left, bh, bw, kind = 0, 0, h, w, 5
            L = layers[k]
            L.up, L.left, L.height, L.width = up, left, bh, bw
            color = tuple(int(v) for v in rng.integers(0, 256, 3))
            for c in range(3):
                L.value_const[c] = color[c]
            alpha = mask = value = None
            a, mode = 1.0, 0
            if kind == 0:
                alpha = (rng.random((bh, bw), dtype=np.float32) * (rng.random((bh, bw)) < 0.5)).astype(np.float32)
                d = ctx.malloc(alpha.nbytes); ctx.upload(d, alpha); keep.append(d)
                L.alpha, L.alpha_stride_el = d, bw
            elif kind == 1:
                mask = (rng.random((bh, bw)) < 0.4).astype(np.uint8)
                d = ctx.malloc(mask.nbytes); ctx.upload(d, mask); keep.append(d)
                L.mask, L.mask_stride = d, bw
                a = float(rng.choice([1.0, 0.35]))
            elif kind == 2:
                value = rng.integers(0, 256, (bh, bw, 3), dtype=np.uint8)
                d = ctx.malloc(value.nbytes); ctx.upload(d, value); keep.append(d)
                L.value, L.value_stride = d, bw * 3
                a = float(rng.choice([1.0, 0.6]))
            elif kind == 3:
                mode = int(rng.choice([1, 2]))
                if rng.random() < 0.5:
                    value = rng.integers(0, 256, (bh, bw, 3), dtype=np.uint8)
                    d = ctx.malloc(value.nbytes); ctx.upload(d, value); keep.append(d)
                    L.value, L.value_stride = d, bw * 3
            elif kind == 4:
                a = float(rng.choice([0.6, 0.0, float(rng.random())]))
            L.alpha_scalar, L.mode = a, mode
            specs.append((p, (up, left, bh, bw), value if value is not None else color, alpha, mask, a, mode))
            k += 1
        begin.append(k)
    d_pages = [ctx.malloc(pg.nbytes) for pg in pages]
    for dp, pg in zip(d_pages, pages):
        ctx.upload(dp, pg)
    ptrs = (ctypes.c_void_p * n_pages)(*d_pages)
    b = np.asarray(begin, np.int32)
    N.check(lib.vkx_fill_u8_batch_dev(ctx.handle, ptrs, n_pages, h, w, cn, w * cn, layers, b.ctypes.data))
    for p in range(n_pages):
        got = np.empty_like(pages[p])
        ctx.download(d_pages[p], got); ctx.sync()
        want = pages[p].copy()
        for (pp, box, value, alpha, mask, a, mode) in specs:
            if pp == p:
                O.fill(want, box, value, mask=mask, alpha=alpha if alpha is not None else a, mode=mode)
        if not np.array_equal(got, want):
            n_bad += 1
            print('MISMATCH', h, w, n_pages, counts, p, int((got != want).sum()), flush=True)
    for d in keep + d_pages:
        ctx.free(d)
    n_batches += 1
    n_layers += k
print(json.dumps({'soak10': 'ok' if n_bad == 0 else 'MISMATCH', 'run': os.environ.get('VKX_RGB_RUN', 'auto'), 'batches': n_batches, 'layers': n_layers,
                  'mismatching_pages': n_bad, 'seconds': round(time.time() - t0, 1)}))
sys.exit(1 if n_bad else 0)
