import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from numpy.random import default_rng
from vkit_amd import _native as N
ctx = N.default_ctx()
img = default_rng(1).integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
for name, fn in (('speckle', lambda r: N.np_speckle_noise(img, 0.2, r)), ('gaussion', lambda r: N.np_gaussion_noise(img, 10.0, r))):
    fn(default_rng(0))
    ctx.set_timing(True); ctx.reset_timings()
    t0 = time.perf_counter()
    for k in range(5):
        fn(default_rng(k))
    dt = (time.perf_counter() - t0) / 5
    print(name, round(dt * 1e3, 3), 'ms', {n: round(v[0] / 5, 3) for n, v in ctx.timings().items()})
    ctx.set_timing(False)
