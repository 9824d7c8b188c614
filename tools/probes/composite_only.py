#!/usr/bin/env python3
"""The batched composite of C4 alone (64 pages of 1024^2, background + 64 text-line layers each): k_composite_rgb without the chain's
setup kernels under it.  Usage: tools/probes/composite_only.py [pages]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from vkit_amd import _native as N
from vkit_amd.batch import ChainBatch
from types import SimpleNamespace

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
size, n_layers, lh, lw = 1024, 64, 32, 512
ctx = N.Context(int(os.environ.get('VKX_DEVICE', 0)))
batch = ChainBatch(ctx)
blank = np.zeros((size, size, 3), np.uint8)
ys = list(range(0, size, 64)) + [size - 1]
sv = np.array([[(x, y) for x in ys] for y in ys], np.int32)
state = SimpleNamespace(result_shape=(size, size), src_image_grid=SimpleNamespace(vertices=sv), dst_image_grid=SimpleNamespace(vertices=sv))
for i in range(B):
    batch.add(blank, state, blur_sigma=None, hue_delta=None, noise=None)
host_layers = [N.make_layer((0, 0, size, size), 3, (200, 200, 200))]
rng2 = np.random.default_rng(0)
for i in range(n_layers):
    alpha = (rng2.random((lh, lw), dtype=np.float32) * (rng2.random((lh, lw)) < 0.3)).astype(np.float32)
    up, left = int(rng2.integers(0, size - lh)), int(rng2.integers(0, size - lw))
    host_layers.append(N.make_layer((up, left, lh, lw), 3, (10, 20, 30), alpha=alpha))
for i in range(B):
    batch.set_layers(i, host_layers)
batch.run(); ctx.sync()
ctx.set_timing(True)
for _ in range(3):
    batch._composite()
ctx.sync(); ctx.reset_timings()
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    batch._composite()
ctx.sync()
dt = (time.perf_counter() - t0) / reps
k = {n: round(v[0] / v[1], 4) for n, v in ctx.timings().items()}
touched = B * (size * size * 3 + n_layers * lh * lw * 4)
print(json.dumps({'pages': B, 'wall_ms_per_call': round(dt * 1e3, 3), 'kernel_ms': k, 'GBps_dst_plus_alpha': round(touched / (k.get('k_composite_rgb', 1e9) * 1e-3) / 1e9, 1),
                  'checksum': int(np.asarray(batch.source(0)).astype(np.int64).sum())}))
