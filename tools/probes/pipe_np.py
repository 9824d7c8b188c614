"""Where the time of HostPipeline.submit_chain(noise_rng=...) goes: device work per image, alone and in the pipeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from numpy.random import default_rng
from vkit_amd import _native as N
from vkit_amd.hostpipe import HostPipeline
from vkit_amd.mechanism.distortion_policy.geometric import mls as P_mls
from vkit_amd.mechanism import distortion as D

ctx = N.default_ctx()
S = 2048
n_img = 8
rng = default_rng(0)
pinned = []
for i in range(n_img):
    a = ctx.pinned_empty((S, S, 3), np.uint8)
    a[...] = rng.integers(0, 256, (S, S, 3), dtype=np.uint8)
    pinned.append(a)
gen = P_mls.SimilarityMlsConfigGenerator(P_mls.SimilarityMlsConfigGeneratorConfig(), 5)
configs = [gen((S, S), default_rng(i)) for i in range(n_img)]
states = [D.similarity_mls.generate_state(c, (S, S)) for c in configs]

# device-only: one plane's stream drawn on the device, repeatedly, synchronised each time / not
dst = ctx.dev_empty((S, S, 3), np.int16)
for sync_each in (True, False):
    ctx.sync()
    t0 = time.perf_counter()
    n = 50
    for k in range(n):
        stream = N.np_stream(default_rng(k))
        jobs = (N.VkxNpJob * 1)(N.np_job(N.NP_NORMAL_I16, stream, S * S * 3, 10.0, dst=dst.ptr))
        res = N.NpResults(ctx, 1) if k == 0 else res
        N.check(N.lib().vkx_np_draw_batch_dev(ctx.handle, jobs, 1, res.array))
        if sync_each:
            ctx.sync()
    ctx.sync()
    print('np draw of one 2048^2x3 plane, sync each' if sync_each else 'np draw, queued', round((time.perf_counter() - t0) / n * 1e3, 3), 'ms')
ctx.set_timing(True)
for k in range(5):
    N.check(N.lib().vkx_np_draw_batch_dev(ctx.handle, jobs, 1, res.array))
ctx.sync()
print(ctx.timings())
ctx.set_timing(False)

for depth in (1, 8):
    for mode in ('remap', 'device_noise', 'numpy_stream'):
        pipe = HostPipeline(ctx, depth=depth)
        def submit(i):
            if mode == 'remap':
                return pipe.submit_remap([pinned[i % n_img]], states[i % n_img])
            if mode == 'device_noise':
                return pipe.submit_chain(pinned[i % n_img], states[i % n_img], blur_sigma=1.2, hue_delta=7, noise_std=10.0, noise_seed=i)
            return pipe.submit_chain(pinned[i % n_img], states[i % n_img], blur_sigma=1.2, hue_delta=7, noise_std=10.0, noise_rng=default_rng(i))
        tickets = [submit(i) for i in range(depth)]
        for t in tickets:
            pipe.result(t)
        n = 48
        t0 = time.perf_counter()
        tickets = []
        ts = tr = 0.0
        for i in range(n):
            a = time.perf_counter()
            tickets.append(submit(i))
            b = time.perf_counter()
            ts += b - a
            if len(tickets) >= depth:
                pipe.result(tickets.pop(0))
                tr += time.perf_counter() - b
        for t in tickets:
            pipe.result(t)
        dt = (time.perf_counter() - t0) / n * 1e3
        print(f'depth {depth} {mode:13s} {dt:.3f} ms/image   in submit {ts / n * 1e3:.3f}, in result {tr / n * 1e3:.3f}')
        pipe.drain()
        pipe.close()
