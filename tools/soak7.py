#!/usr/bin/env python3
"""Seventh soak: the numpy streams of round 3 against numpy itself on random inputs -- calls of 1 .. 150 ragged streams (through
the inline-job form, the ring form and the chunk pipeline), every kind (int16 plane, in-place add, speckle, choice, impulse),
page-locked and pageable results, the generator state after the call; ChainBatch / HostPipeline with ``noise_rng`` against the
oracle chain fed numpy's plane.  Usage: tools/soak7.py <seconds> <seed>"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from numpy.random import default_rng

import oracle as O
from vkit_amd import _native as N
from vkit_amd.batch import ChainBatch
from vkit_amd.hostpipe import HostPipeline

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = N.default_ctx()
t0 = time.time()
counts = {'calls': 0, 'streams': 0, 'compared': 0, 'flagged': 0, 'samples': 0, 'chains': 0}
pipe = HostPipeline(depth=4, lanes=4)


def state_after(seed, draws):
    return N.pcg64_jump(*N.np_stream(default_rng(seed)), draws)


while time.time() - t0 < budget:
    if rng.random() < 0.85:
        B = int(rng.choice([1, 2, 7, 8, 9, 40, 63, 64, 65, 97, 150]))
        kind = int(rng.choice([N.NP_NORMAL_I16, N.NP_NORMAL_ADD_U8, N.NP_SPECKLE_U8, N.NP_CHOICE3_U8, N.NP_IMPULSE_U8, N.NP_NORMAL_TILES]))
        big = 400_000 if B <= 9 else 60_000
        sizes = [int(v) for v in rng.integers(1, big, B)]
        seeds = [int(v) for v in rng.integers(0, 2 ** 62, B)]
        stds = [float(v) for v in (rng.uniform(0.01, 1.5, B) if kind == N.NP_SPECKLE_U8 else rng.uniform(0.3, 60.0, B))]
        pinned = rng.random() < 0.5
        res = N.NpResults(ctx, B) if pinned else None
        res_array = res.array if pinned else (N.VkxNpResult * B)()
        jobs = (N.VkxNpJob * B)()
        outs, srcs, cdfs, cns = [], [], [], []
        for i in range(B):
            stream = N.np_stream(default_rng(seeds[i]))
            if kind == N.NP_NORMAL_I16:
                d = ctx.dev_empty((sizes[i],), np.int16)
                jobs[i] = N.np_job(kind, stream, sizes[i], stds[i], dst=d.ptr)
            elif kind == N.NP_NORMAL_TILES:
                d = ctx.dev_empty((N.np_tiles_layout(sizes[i])[4],), np.uint8)
                jobs[i] = N.np_job(kind, stream, sizes[i], stds[i], dst=d.ptr)
            elif kind in (N.NP_NORMAL_ADD_U8, N.NP_SPECKLE_U8):
                px = default_rng(seeds[i] + 1).integers(0, 256, sizes[i], dtype=np.uint8)
                srcs.append(px)
                s = ctx.to_device(px)
                d = s if kind == N.NP_NORMAL_ADD_U8 else ctx.dev_empty((sizes[i],), np.uint8)
                outs.append(s)
                jobs[i] = N.np_job(kind, stream, sizes[i], stds[i], src=s.ptr, dst=d.ptr)
            else:
                p = rng.dirichlet([8.0, 1.0, 1.0])
                cdf = N._choice_cdf(p)
                cdfs.append((p, cdf))
                cn = int(rng.integers(1, 5)) if kind == N.NP_IMPULSE_U8 else 1
                cns.append(cn)
                if kind == N.NP_IMPULSE_U8:
                    px = default_rng(seeds[i] + 1).integers(0, 256, (sizes[i], cn), dtype=np.uint8)
                    srcs.append(px)
                    s = ctx.to_device(px)
                    outs.append(s)
                    d = ctx.dev_empty((sizes[i], cn), np.uint8)
                    jobs[i] = N.np_job(kind, stream, sizes[i], 0.0, cdf=cdf, cn=cn, src=s.ptr, dst=d.ptr)
                else:
                    d = ctx.dev_empty((sizes[i],), np.uint8)
                    jobs[i] = N.np_job(kind, stream, sizes[i], 0.0, cdf=cdf, dst=d.ptr)
            outs.append(d)
            if kind in (N.NP_NORMAL_I16, N.NP_CHOICE3_U8, N.NP_NORMAL_TILES):
                srcs.append(None)
        N.check(N.lib().vkx_np_draw_batch_dev(ctx.handle, jobs, B, res_array))
        ctx.sync()
        k = 0
        for i in range(B):
            if res_array[i].flags:           # a decision inside the libm margin: the host would draw; nothing to compare
                counts['flagged'] += 1
                continue
            counts['compared'] += 1
            ref = default_rng(seeds[i])
            if kind == N.NP_NORMAL_I16:
                got = [o for o in outs if o.ptr == jobs[i].dst][0].host()
                assert (got == np.round(ref.normal(0, stds[i], sizes[i])).astype(np.int16)).all(), ('i16', B, i)
            elif kind == N.NP_NORMAL_TILES:
                buf = [o for o in outs if o.ptr == jobs[i].dst][0]
                want = np.round(ref.normal(0, stds[i], sizes[i])).astype(np.int16)
                assert (N.np_tiles_plane(buf.host(), sizes[i]) == want).all(), ('tiles', B, i)
                plane = ctx.dev_empty((sizes[i],), np.int16)
                N.check(N.lib().vkx_np_tiles_expand_dev(ctx.handle, ctypes.c_void_p(buf.ptr), sizes[i], ctypes.c_void_p(plane.ptr)))
                assert (plane.host() == want).all(), ('tiles expand', B, i)
            elif kind == N.NP_NORMAL_ADD_U8:
                got = [o for o in outs if o.ptr == jobs[i].dst][0].host()
                noise = np.round(ref.normal(0, stds[i], sizes[i])).astype(np.int16)
                assert (got == np.clip(srcs[i].astype(np.int16) + noise, 0, 255).astype(np.uint8)).all(), ('add', B, i)
            elif kind == N.NP_SPECKLE_U8:
                got = [o for o in outs if o.ptr == jobs[i].dst][0].host()
                noise = ref.normal(0, stds[i], sizes[i])
                want = np.clip(srcs[i] + srcs[i] * noise, 0, 255).astype(np.uint8)
                assert (got == want).all(), ('speckle', B, i)
            elif kind == N.NP_CHOICE3_U8:
                got = [o for o in outs if o.ptr == jobs[i].dst][0].host()
                assert (got == ref.choice((0, 1, 2), size=sizes[i], p=cdfs[i][0])).all(), ('choice', B, i)
            else:
                got = [o for o in outs if o.ptr == jobs[i].dst][0].host()
                sel = ref.choice((0, 1, 2), size=sizes[i], p=cdfs[i][0])
                want = srcs[i].copy()
                want[sel == 1] = 255
                want[sel == 2] = 0
                assert (got == want).all(), ('impulse', B, i)
            assert state_after(seeds[i], res_array[i].draws) == ref.bit_generator.state['state']['state'], ('state', kind, B, i)
            counts['samples'] += sizes[i]
        counts['calls'] += 1
        counts['streams'] += B
    else:
        from test_gpu_hostpipe import _state, synthetic_grid
        n_img = int(rng.integers(1, 5))
        mode = str(rng.choice(['tiles', 'tiles', 'planes', 'late']))
        batch = ChainBatch(stream_noise_mode=mode)
        big = rng.random() < 0.25            # pages of several generator tiles per row band: rows that straddle two slots
        from vkit_amd.mechanism.distortion.photometric.streak import LineStreakConfig
        wants = []
        tickets = []
        for k in range(n_img):
            h, w = (int(rng.integers(500, 1300)), int(rng.integers(700, 1500))) if big else (int(rng.integers(60, 400)), int(rng.integers(60, 500)))
            sv, dv, dshape = synthetic_grid(h, w, int(rng.integers(8, 40)), float(rng.uniform(1, 9)), seed=int(rng.integers(1 << 30)))
            st = _state(sv, dv, dshape)
            image = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            std, seed = float(rng.uniform(1, 40)), int(rng.integers(1 << 40))
            mx, my = O.grid_to_map(sv, dv, dshape)
            base = O.color_shift_rgb(O.gaussian_blur(O.remap(image, mx, my), 5, 1.0), 37)
            plane = np.round(default_rng(seed).normal(0, std, tuple(dshape) + (3,))).astype(np.int16)
            want = O.add_noise_i16(base, plane)
            streak = None
            if rng.random() < 0.3:
                streak = LineStreakConfig(thickness=int(rng.integers(1, 4)), gap=int(rng.integers(3, 20)), alpha=float(rng.uniform(0.1, 0.9)),
                                          color=tuple(int(v) for v in rng.integers(0, 256, 3)), enable_vert=True, enable_hori=bool(rng.random() < 0.5))
                want = O.line_streak(want, streak.thickness, streak.gap, streak.dash_thickness, streak.dash_gap, streak.color, streak.alpha,
                                     streak.enable_vert, streak.enable_hori)
            wants.append(want)
            batch.add(image, st, blur_sigma=1.0, hue_delta=37, noise_std=std, noise_rng=default_rng(seed), streak=streak)
            tickets.append(pipe.submit_chain(image, st, blur_sigma=1.0, hue_delta=37, noise_std=std, noise_rng=default_rng(seed), streak=streak))
        batch.run()
        for k in range(n_img):
            assert (batch.result(k) == wants[k]).all(), ('batch', mode, k)
            assert (pipe.result(tickets[k])[0] == wants[k]).all(), ('pipeline', k)
        batch.close()
        counts['chains'] += n_img
pipe.close()
print('soak7 ok', counts, round(time.time() - t0), 's')
