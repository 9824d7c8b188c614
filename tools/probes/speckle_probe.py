import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from numpy.random import default_rng
from vkit_amd import _native as N
ctx = N.default_ctx()
P = 1024
img = default_rng(100).integers(0, 256, (P, P, 3), dtype=np.uint8)
for std in (0.05, 0.2, 0.5):
    for k in range(3):
        rng = default_rng(k)
        t0 = time.perf_counter()
        out = N.np_speckle_noise(img, std, rng)
        ctx.sync()
        print('std', std, 'device result', out is not None, round((time.perf_counter() - t0) * 1e3, 3), 'ms')
ctx.set_timing(True)
out = N.np_speckle_noise(img, 0.2, default_rng(5)); ctx.sync()
print(ctx.timings())
