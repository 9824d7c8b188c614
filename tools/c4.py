#!/usr/bin/env python3
"""C4-shaped page composite timing (device resident): 64 text-line score-map layers of 32 x 512 on a 1024^2 RGB page."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time

import numpy as np

from vkit_amd import _native as N

ctx = N.Context(0)
rng = np.random.default_rng(0)
size, n_layers, lh, lw = 1024, 64, 32, 512
page = np.full((size, size, 3), 200, np.uint8)
d_page = ctx.malloc(page.nbytes)
layers = (N.VkxLayer * n_layers)()
keep = []
for i in range(n_layers):
    alpha = (rng.random((lh, lw), dtype=np.float32) * (rng.random((lh, lw)) < 0.3)).astype(np.float32)
    d_alpha = ctx.malloc(alpha.nbytes)
    ctx.upload(d_alpha, alpha)
    keep.append(d_alpha)
    L = layers[i]
    L.up, L.left, L.height, L.width = int(rng.integers(0, size - lh)), int(rng.integers(0, size - lw)), lh, lw
    L.alpha, L.alpha_stride_el, L.alpha_scalar = d_alpha, lw, 1.0
    L.value_const[0], L.value_const[1], L.value_const[2] = 10, 20, 30
lib = N.lib()
ctx.set_timing(True)
res = {}
for label, count in (('one_call_64_layers', n_layers), ('per_layer_calls', 1)):
    times = []
    for rep in range(6):
        ctx.upload(d_page, page)
        ctx.sync()
        t0 = time.perf_counter()
        if count == n_layers:
            N.check(lib.vkx_fill_u8_dev(ctx.handle, d_page, size, size, 3, size * 3, layers, n_layers))
        else:
            for i in range(n_layers):
                N.check(lib.vkx_fill_u8_dev(ctx.handle, d_page, size, size, 3, size * 3,
                                            ctypes.cast(ctypes.byref(layers[i]), ctypes.POINTER(N.VkxLayer)), 1))
        ctx.sync()
        times.append(time.perf_counter() - t0)
    out = np.empty_like(page)
    ctx.download(d_page, out)
    res[label] = {'wall_ms_min': min(times) * 1e3, 'checksum': int(out.astype(np.int64).sum())}
res['kernel_ms'] = {k: [round(v[0], 4), v[1]] for k, v in ctx.timings().items()}
print(json.dumps(res))
