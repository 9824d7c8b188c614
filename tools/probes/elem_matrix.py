#!/usr/bin/env python3
"""vkx_grid_remap (element mode of the tile kernel) over sizes x element sets: kernel time of k_tile_remap.
Usage: [VKX_TILE_NW=4|8] elem_matrix.py [sizes,comma,separated] [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from numpy.random import default_rng

from vkit_amd import _native as N
from vkit_amd.mechanism import distortion as D
from vkit_amd.mechanism.distortion_policy.geometric import mls as P_mls

ctx = N.Context(int(os.environ.get('VKX_DEVICE', 0)))
lib = N.lib()
out = {}
SIZES = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [512, 1024, 2048, 4096]
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for size in SIZES:
    gen = P_mls.SimilarityMlsConfigGenerator(P_mls.SimilarityMlsConfigGeneratorConfig(), 5)
    state = D.similarity_mls.generate_state(gen((size, size), default_rng(0)), (size, size))
    dh, dw = state.result_shape
    sv = np.ascontiguousarray(state.src_image_grid.vertices, np.int32)
    dv = np.ascontiguousarray(state.dst_image_grid.vertices, np.int32)
    rng = default_rng(1)
    image = rng.integers(0, 256, (size, size, 3), dtype=np.uint8)
    mask = (rng.random((size, size)) < 0.5).astype(np.uint8)
    score = rng.random((size, size), dtype=np.float32)
    for label, srcs in (('image', [image]), ('mask', [mask]), ('image+mask', [image, mask]), ('image+mask+score', [image, mask, score])):
        elems = (N.VkxElem * len(srcs))()
        keep = []
        for i, a in enumerate(srcs):
            d_src = ctx.malloc(a.nbytes); ctx.upload(d_src, a)
            cn = 3 if a.ndim == 3 else 1
            isf = a.dtype == np.float32
            d_dst = ctx.malloc(dh * dw * cn * a.itemsize)
            keep += [d_src, d_dst]
            elems[i].src, elems[i].dst = d_src, d_dst
            elems[i].src_stride = size * cn if not isf else size
            elems[i].dst_stride = dw * cn if not isf else dw
            elems[i].cn, elems[i].is_f32 = cn, int(isf)
        d_sv = ctx.malloc(sv.nbytes); ctx.upload(d_sv, sv)
        d_dv = ctx.malloc(dv.nbytes); ctx.upload(d_dv, dv)
        keep += [d_sv, d_dv]
        def call():
            N.check(lib.vkx_grid_remap_dev(ctx.handle, elems, len(srcs), size, size, d_sv, d_dv, sv.shape[0], sv.shape[1], dh, dw))
        for _ in range(3):
            call()
        ctx.sync(); ctx.set_timing(True); ctx.reset_timings()
        for _ in range(REPS):
            call()
        ctx.sync()
        t = ctx.timings(); ctx.set_timing(False)
        out[f'{size}:{label}'] = round(t['k_tile_remap'][0] / t['k_tile_remap'][1], 4)
        for d in keep:
            ctx.free(d)
print(json.dumps(out))
