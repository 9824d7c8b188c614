#!/usr/bin/env python3
"""C4-shaped page synthesis timing (device resident): 64 text-line score-map layers of 32 x 512 composited onto a 1024^2 RGB
page (one call / per-layer calls), and -- BASELINE config 3 -- a batch of 64 such pages composited and sent through the full
distortion chain (camera_cubic_curve remap + gaussian_blur + color_shift + gaussion_noise) without leaving HBM."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time

import numpy as np

from vkit_amd import _native as N

ctx = N.Context(int(os.environ.get('VKX_DEVICE', 0)))
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

# ---- page synthesis, resident: B pages, each = background fill + 64 text layers in one composite call, then ONE chain launch
from numpy.random import default_rng
from vkit_amd.batch import ChainBatch
from vkit_amd.mechanism import distortion as D
from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam

B = 64
gen = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), 5)
batch = ChainBatch(ctx)
blank = np.zeros((size, size, 3), np.uint8)
for i in range(B):
    state = D.camera_cubic_curve.generate_state(gen((size, size), default_rng(i)), (size, size))
    noise = np.round(default_rng(900 + i).normal(0, 10.0, tuple(state.result_shape) + (3,))).astype(np.int16)
    batch.add(blank, state, blur_sigma=1.0, hue_delta=37, noise=noise)

# the same 65 layers on every page, through the product API: ChainBatch.set_layers uploads the planes, every run composites all
# pages with ONE vkx_fill_u8_batch_dev launch and then issues the chain
host_layers = [N.make_layer((0, 0, size, size), 3, (200, 200, 200))]
rng2 = np.random.default_rng(0)
for i in range(n_layers):
    alpha = (rng2.random((lh, lw), dtype=np.float32) * (rng2.random((lh, lw)) < 0.3)).astype(np.float32)
    up, left = int(rng2.integers(0, size - lh)), int(rng2.integers(0, size - lw))
    host_layers.append(N.make_layer((up, left, lh, lw), 3, (10, 20, 30), alpha=alpha))
for i in range(B):
    batch.set_layers(i, host_layers)


def synth_step():
    batch.run()


synth_step(); ctx.sync()
ctx.reset_timings()
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    synth_step()
ctx.sync()
dt = (time.perf_counter() - t0) / reps
k = {n: round(v[0] / reps, 4) for n, v in ctx.timings().items()}
# the same runs without the per-kernel event pairs (a pair costs a few microseconds of stream time; five kernels per batch)
ctx.set_timing(False)
synth_step(); ctx.sync()
t0 = time.perf_counter()
for _ in range(reps * 3):
    synth_step()
ctx.sync()
dt_plain = (time.perf_counter() - t0) / (reps * 3)
# the composite of page 0 is the single-page result above; the chain on it is verified by the GPU test suite
first = batch.source(0)
res['page_synth_resident'] = {'pages': B, 'composite': 'ChainBatch.set_layers: one batched launch per run', 'ms_per_batch': round(dt_plain * 1e3, 3), 'ms_per_batch_with_kernel_events': round(dt * 1e3, 3), 'pages_per_s': round(B / dt_plain), 'Mpx_s': round(B * size * size / dt_plain / 1e6),
                              'kernel_ms_per_batch': k, 'composite_matches_single_page': bool((first == out).all())}
print(json.dumps(res))
