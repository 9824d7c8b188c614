#!/usr/bin/env python3
"""Device-resident timings of the rows around the chain: page resizing (every sampled interpolation on a 2048^2 page
image, mask and score map, shrink to 1229 x 1229 and growth to 2458 x 2458), label painting is in tools/kernels.py,
and point projection of 5000 points (host arrays in and out, the call the pipeline makes)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from numpy.random import default_rng

from vkit_amd import _native as N

ctx = N.Context(0)
lib = N.lib()
rng = default_rng(0)
S = 2048
img = rng.integers(0, 256, (S, S, 3), dtype=np.uint8)
mask = ((rng.random((S, S)) < 0.3) * 255).astype(np.uint8)
score = rng.random((S, S), dtype=np.float32)
d_img = ctx.malloc(img.nbytes); ctx.upload(d_img, img)
d_mask = ctx.malloc(mask.nbytes); ctx.upload(d_mask, mask)
d_score = ctx.malloc(score.nbytes); ctx.upload(d_score, score)
names = ['NEAREST', 'LINEAR', 'CUBIC', 'AREA', 'LANCZOS4', 'LINEAR_EXACT', 'NEAREST_EXACT']
res = {'resize_2048_page': {}}
for D in (1229, 2458):
    d_o3 = ctx.malloc(D * D * 3); d_o1 = ctx.malloc(D * D); d_of = ctx.malloc(D * D * 4)
    for inter in (6, 5, 2, 4, 3):
        if inter == 3 and D > S:
            continue
        def run():
            N.check(lib.vkx_resize_u8_dev(ctx.handle, d_img, S, S, 3, S * 3, d_o3, D, D, D * 3, inter))
            N.check(lib.vkx_resize_u8_dev(ctx.handle, d_mask, S, S, 1, S, d_o1, D, D, D, inter))
            N.check(lib.vkx_resize_f32_dev(ctx.handle, d_score, S, S, S, d_of, D, D, D, inter))
        run(); ctx.sync()
        ts = []
        for _ in range(5):
            ctx.sync(); t0 = time.perf_counter(); run(); ctx.sync(); ts.append(time.perf_counter() - t0)
        ctx.set_timing(True); ctx.reset_timings()
        for _ in range(3):
            run()
        ctx.sync()
        kern = sum(v[0] for v in ctx.timings().values()) / 3
        ctx.set_timing(False); ctx.reset_timings()
        bytes_moved = (S * S + D * D) * (3 + 1 + 4)
        res['resize_2048_page'][f'{names[inter]}_to_{D}'] = {'ms_image_mask_score': round(min(ts) * 1e3, 3),
                                                            'kernels_ms': round(kern, 3),
                                                            'GBps_on_S_plus_D': round(bytes_moved / min(ts) / 1e9)}
# point projection
import vkit_amd.mechanism.distortion as Dm
from vkit_amd.mechanism.distortion_policy.geometric import mls as P_mls
cfg = P_mls.SimilarityMlsConfigGenerator(P_mls.SimilarityMlsConfigGeneratorConfig(), 5)((S, S), default_rng(0))
state = Dm.similarity_mls.generate_state(cfg, (S, S))
sv, dv = state.src_image_grid.vertices, state.dst_image_grid.vertices
gs = state.src_image_grid.grid_size
pts = rng.integers(0, S - 1, (5000, 2)).astype(np.int32)
smooth = pts.astype(np.float64)
N.project_points(sv, dv, gs, pts, smooth)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); N.project_points(sv, dv, gs, pts, smooth); ts.append(time.perf_counter() - t0)
res['project_5000_points_host_call_ms'] = round(min(ts) * 1e3, 3)
from vkit_amd.element import Point
from vkit_amd.mechanism.distortion.geometric.grid_rendering.interface import FuncImageGridBased
t0 = time.perf_counter()
for x, y in pts[:200]:
    FuncImageGridBased.func_point(cfg, state, (S, S), Point.create(y=int(y), x=int(x)), None)
res['func_point_host_loop_ms_per_5000'] = round((time.perf_counter() - t0) / 200 * 5000 * 1e3, 1)
print(json.dumps(res))
