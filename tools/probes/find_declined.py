import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import numpy as np
from numpy.random import default_rng
from vkit_amd import _native as N
import importlib.util
src = open('/root/repo/tools/soak8.py').read().split("g = default_rng(2024)")[0]
ns = {'__file__': '/root/repo/tools/soak8.py'}
exec(compile(src.replace("IMAGES = int(sys.argv[1]) if len(sys.argv) > 1 else 120", "IMAGES = 3000"), 'soak8_head', 'exec'), ns)
image = ns['image']
g = default_rng(2024)
for k in range(3000):
    img = image(g, k)
    seed, skip = int(g.integers(1 << 30)), int(g.integers(0, 1000))
    r = default_rng(seed); r.random(skip)
    got = N.np_poisson_u8(img, r)
    if got is None:
        print('declined', k, img.shape, 'kind', k % 7, 'flags', N.np_poisson_last_flags(), 'seed', seed, 'skip', skip, 'min/max', int(img.min()), int(img.max()))
        # repeat: deterministic?
        r2 = default_rng(seed); r2.random(skip)
        print('again', N.np_poisson_u8(img, r2) is None, N.np_poisson_last_flags())
        break
