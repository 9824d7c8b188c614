import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from vkit_amd import _native as N
ctx = N.default_ctx()
img = np.random.default_rng(1).integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
for std in (0.05, 0.2, 0.3):
    for it in range(3):
        rng = np.random.default_rng(it)
        ctx.set_timing(True); ctx.reset_timings()
        t = time.perf_counter(); out = N.np_speckle_noise(img, std, rng, ctx); dt = time.perf_counter() - t
        print(std, it, 'None' if out is None else 'ok', round(dt * 1e3, 2), {k: round(v[0], 3) for k, v in ctx.timings().items()})
for it in range(3):
    rng = np.random.default_rng(it)
    t = time.perf_counter(); out = N.np_gaussion_noise(img, 10.0, rng, ctx); dt = time.perf_counter() - t
    print('gauss', it, 'None' if out is None else 'ok', round(dt * 1e3, 2))
