#!/usr/bin/env python3
"""The drop-in measured end to end: host arrays in, host arrays out (PCIe inclusive), next to the device-resident rate of
bench.py.  Never the bench `value`; bench.py embeds this as `dropin`.

  sync_mls_2048        similarity_mls.distort(config, image=...) one call at a time: state construction (device lattice
                       kernel), upload, tile kernel, download into a recycled page-locked result
  pipeline_remap_2048  the same remaps through HostPipeline (8 lanes): uploads / kernels / downloads of different jobs
                       overlap on the copy streams; inputs in page-locked memory, states built beforehand
  pipeline_chain_2048  C3's chain per image through HostPipeline, with its int16 noise plane uploaded (6 B / result px),
                       and with the plane drawn on the device instead (throughput mode)
  random_distortion_1024  RandomDistortion.distort on 1024^2 pages (C4's distortion step), default policy table; its time
                       is the host arithmetic of the rng-stream members (tools/rd_profile.py)
Usage: tools/dropin.py [out.json] [--quick]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from numpy.random import default_rng


def measure(quick=False):
    from vkit_amd import _native as N
    from vkit_amd.element import Image
    from vkit_amd.hostpipe import HostPipeline
    from vkit_amd.mechanism import distortion as D
    from vkit_amd.mechanism.distortion_policy import random_distortion_factory
    from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam, mls as P_mls
    ctx = N.default_ctx()
    S = 2048
    out = {}
    n_img = 6 if quick else 12
    rng = default_rng(0)
    pinned = []
    for i in range(n_img):
        a = ctx.pinned_empty((S, S, 3), np.uint8)
        a[...] = rng.integers(0, 256, (S, S, 3), dtype=np.uint8)
        pinned.append(a)
    gen = P_mls.SimilarityMlsConfigGenerator(P_mls.SimilarityMlsConfigGeneratorConfig(), 5)
    configs = [gen((S, S), default_rng(i)) for i in range(n_img)]

    # ---- synchronous operator calls
    images = [Image(mat=np.array(p)) for p in pinned]           # pageable copies: what a caller normally holds
    D.similarity_mls.distort(configs[0], image=images[0])
    reps = 2 if quick else 5
    t0 = time.perf_counter()
    for r in range(reps):
        for cfg, im in zip(configs, images):
            res = D.similarity_mls.distort(cfg, image=im)
    dt = time.perf_counter() - t0
    n = reps * n_img
    out['sync_mls_2048'] = {'images_per_s': n / dt, 'mpx_per_s': n * S * S / dt / 1e6, 'ms_per_image': dt / n * 1e3,
                            'note': 'similarity_mls.distort(config, image=Image) per call, pageable input, state built in the call'}
    t0 = time.perf_counter()
    for cfg in configs:
        D.similarity_mls.generate_state(cfg, (S, S))
    out['sync_mls_2048']['state_ms'] = (time.perf_counter() - t0) / n_img * 1e3

    # ---- overlapped pipeline, remap only
    states = [D.similarity_mls.generate_state(cfg, (S, S)) for cfg in configs]
    jobs = (8 if quick else 30) * n_img
    for depth in (1, 8):
        with HostPipeline(ctx, depth=depth) as pipe:
            for k in range(depth):
                pipe.submit_remap([pinned[k % n_img]], states[k % n_img])
            pipe.drain()
            t0 = time.perf_counter()
            tickets = []
            chk = 0
            for k in range(jobs):
                tickets.append(pipe.submit_remap([pinned[k % n_img]], states[k % n_img]))
                if k >= depth - 1:
                    chk += int(pipe.result(tickets[k - depth + 1])[0][7, 7, 1])
            pipe.drain()
            dt = time.perf_counter() - t0
        out[f'pipeline_remap_2048_depth{depth}'] = {
            'images_per_s': jobs / dt, 'mpx_per_s': jobs * S * S / dt / 1e6, 'ms_per_image': dt / jobs * 1e3,
            'link_gb_per_s': jobs * (S * S * 3 + states[0].result_shape[0] * states[0].result_shape[1] * 3) / dt / 1e9,
            'note': f'HostPipeline depth {depth}, page-locked inputs, states prebuilt, results are page-locked views'}

    # ---- overlapped pipeline, C3 chain with the int16 noise plane as an input
    cgen = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), 5)
    cstates = [D.camera_cubic_curve.generate_state(cgen((S, S), default_rng(i)), (S, S)) for i in range(n_img)]
    noises = []
    for i, st in enumerate(cstates):
        nz = ctx.pinned_empty(tuple(st.result_shape) + (3,), np.int16)
        nz[...] = np.round(default_rng(5000 + i).normal(0, 10.0, nz.shape)).astype(np.int16)
        noises.append(nz)
    jobs = (4 if quick else 16) * n_img
    with HostPipeline(ctx) as pipe:
        for _ in pipe.slots:            # every slot allocates its device / page-locked buffers on first use: not timed
            pipe.submit_chain(pinned[0], cstates[0], blur_sigma=1.0, hue_delta=37, noise=noises[0])
        pipe.drain()
        t0 = time.perf_counter()
        tickets = []
        for k in range(jobs):
            i = k % n_img
            tickets.append(pipe.submit_chain(pinned[i], cstates[i], blur_sigma=1.0, hue_delta=37, noise=noises[i]))
            if k >= 7:
                pipe.result(tickets[k - 7])
        pipe.drain()
        dt = time.perf_counter() - t0
    out['pipeline_chain_2048_depth8'] = {'images_per_s': jobs / dt, 'mpx_per_s': jobs * S * S / dt / 1e6,
                                        'ms_per_image': dt / jobs * 1e3,
                                        'note': 'C3 chain per image incl. upload of its int16 noise plane (6 B per result pixel)'}

    # the same chain with the noise plane drawn on the device (throughput mode): only the page crosses the link
    with HostPipeline(ctx) as pipe:
        for _ in pipe.slots:
            pipe.submit_chain(pinned[0], cstates[0], blur_sigma=1.0, hue_delta=37, noise_std=10.0, noise_seed=1)
        pipe.drain()
        t0 = time.perf_counter()
        tickets = []
        for k in range(jobs):
            i = k % n_img
            tickets.append(pipe.submit_chain(pinned[i], cstates[i], blur_sigma=1.0, hue_delta=37, noise_std=10.0, noise_seed=k))
            if k >= 7:
                pipe.result(tickets[k - 7])
        pipe.drain()
        dt = time.perf_counter() - t0
    out['pipeline_chain_2048_device_noise'] = {'images_per_s': jobs / dt, 'mpx_per_s': jobs * S * S / dt / 1e6,
                                               'ms_per_image': dt / jobs * 1e3,
                                               'note': 'C3 chain per image, noise drawn on the device (vkx_noise_normal_i16_dev): '
                                                       'the distribution of the reference, not its numpy values'}

    # the same chain, value-exact: the plane np.round(default_rng(5000 + i).normal(0, 10, shape)) is drawn on the device from
    # the numpy stream (vkx_np_draw_batch_dev), nothing but the page crosses the link; checked against the uploaded-plane job
    with HostPipeline(ctx) as pipe:
        want = pipe.result(pipe.submit_chain(pinned[0], cstates[0], blur_sigma=1.0, hue_delta=37, noise=noises[0]))[0].copy()
        got = pipe.result(pipe.submit_chain(pinned[0], cstates[0], blur_sigma=1.0, hue_delta=37, noise_std=10.0,
                                            noise_rng=default_rng(5000)))[0]
        exact = bool((got == want).all())
        for _ in pipe.slots:
            pipe.submit_chain(pinned[0], cstates[0], blur_sigma=1.0, hue_delta=37, noise_std=10.0, noise_rng=default_rng(5000))
        pipe.drain()
        t0 = time.perf_counter()
        tickets = []
        for k in range(jobs):
            i = k % n_img
            tickets.append(pipe.submit_chain(pinned[i], cstates[i], blur_sigma=1.0, hue_delta=37, noise_std=10.0,
                                             noise_rng=default_rng(5000 + i)))
            if k >= 7:
                pipe.result(tickets[k - 7])
        pipe.drain()
        dt = time.perf_counter() - t0
    out['pipeline_chain_2048_numpy_stream'] = {'images_per_s': jobs / dt, 'mpx_per_s': jobs * S * S / dt / 1e6,
                                               'ms_per_image': dt / jobs * 1e3, 'equals_uploaded_plane_job': exact,
                                               'note': 'C3 chain per image, the numpy noise stream of the image drawn on the device '
                                                       '(value for value the reference\'s plane): only the page crosses the link'}

    # ---- RandomDistortion on 1024^2 pages
    P = 1024
    pages = [Image(mat=default_rng(100 + i).integers(0, 256, (P, P, 3), dtype=np.uint8)) for i in range(8)]
    rd = random_distortion_factory.create()
    rd.distort(default_rng(0), image=pages[0])
    n = 32 if quick else 64
    t0 = time.perf_counter()
    for k in range(n):
        rd.distort(default_rng(k), image=pages[k % len(pages)])
    dt = time.perf_counter() - t0
    out['random_distortion_1024'] = {'pages_per_s': n / dt, 'mpx_per_s': n * P * P / dt / 1e6, 'ms_per_page': dt / n * 1e3,
                                     'note': 'RandomDistortion.distort(rng, image=1024^2 page), default policy table'}
    return out


if __name__ == '__main__':
    res = measure(quick='--quick' in sys.argv)
    print(json.dumps(res, indent=1))
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    if args:
        with open(args[0], 'w') as f:
            json.dump(res, f, indent=1)
