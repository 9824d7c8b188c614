#!/usr/bin/env python3
"""C5 timing: one 4096^2 similarity_mls state applied to Image + Mask + ScoreMap (device resident, generic grid path),
and C2-style single 2048^2 RGB image through the same path."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time

import numpy as np
from numpy.random import default_rng

from vkit_amd import _native as N
from vkit_amd.mechanism import distortion as D
from vkit_amd.mechanism.distortion_policy.geometric import mls as P_mls

ctx = N.Context(int(os.environ.get('VKX_DEVICE', 0)))
lib = N.lib()
res = {}
for size, label in ((4096, 'C5_4096_three_elements'), (2048, 'C2_2048_image_only')):
    gen = P_mls.SimilarityMlsConfigGenerator(P_mls.SimilarityMlsConfigGeneratorConfig(), 5)
    cfg = gen((size, size), default_rng(0))
    t0 = time.perf_counter()
    state = D.similarity_mls.generate_state(cfg, (size, size))
    t_state = time.perf_counter() - t0
    dh, dw = state.result_shape
    sv = np.ascontiguousarray(state.src_image_grid.vertices, np.int32)
    dv = np.ascontiguousarray(state.dst_image_grid.vertices, np.int32)
    rng = default_rng(1)
    image = rng.integers(0, 256, (size, size, 3), dtype=np.uint8)
    mask = (rng.random((size, size)) < 0.5).astype(np.uint8)
    score = rng.random((size, size), dtype=np.float32)
    srcs = [image, mask, score] if size == 4096 else [image]
    elems = (N.VkxElem * len(srcs))()
    for i, a in enumerate(srcs):
        d_src = ctx.malloc(a.nbytes); ctx.upload(d_src, a)
        cn = 3 if a.ndim == 3 else 1
        isf = a.dtype == np.float32
        d_dst = ctx.malloc(dh * dw * cn * a.itemsize)
        elems[i].src, elems[i].dst = d_src, d_dst
        elems[i].src_stride = size * cn if not isf else size
        elems[i].dst_stride = dw * cn if not isf else dw
        elems[i].cn, elems[i].is_f32 = cn, int(isf)
    d_sv = ctx.malloc(sv.nbytes); ctx.upload(d_sv, sv)
    d_dv = ctx.malloc(dv.nbytes); ctx.upload(d_dv, dv)
    ctx.set_timing(True)
    times = []
    for rep in range(5):
        ctx.sync(); t0 = time.perf_counter()
        N.check(lib.vkx_grid_remap_dev(ctx.handle, elems, len(srcs), size, size, d_sv, d_dv, sv.shape[0], sv.shape[1], dh, dw))
        ctx.sync(); times.append(time.perf_counter() - t0)
        if rep == 0:
            ctx.reset_timings()
    k = ctx.timings()
    res[label] = {'state_s': round(t_state, 2), 'result_shape': [dh, dw], 'wall_ms_min': round(min(times) * 1e3, 3),
                  'kernels_ms_per_call': {n: round(v[0] / v[1], 3) for n, v in k.items()},
                  'Mpx_s': round(size * size / min(times) / 1e6)}
    ctx.reset_timings()
print(json.dumps(res))
