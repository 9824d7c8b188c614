import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from numpy.random import default_rng
from vkit_amd import _native as N
g = default_rng(3)
def page(sz):
    p = np.full((sz, sz, 3), 255, np.uint8)
    for k in range(0, sz - 40, 40):
        p[k + 8:k + 24, 32:sz - 32] = g.integers(0, 60, (16, sz - 64, 3), dtype=np.uint8) * (g.random((16, sz - 64, 1)) < 0.4)
    return p
cases = [('page2048', page(2048)), ('uniform4096', g.integers(0, 256, (4096, 4096, 3), dtype=np.uint8)), ('dark2048', g.integers(0, 13, (2048, 2048, 3), dtype=np.uint8)),
         ('mixed', np.concatenate([g.integers(0, 13, (500000,), dtype=np.uint8), np.full(700001, 255, np.uint8), np.zeros(300000, np.uint8), g.integers(0, 256, (900000,), dtype=np.uint8)]))]
for name, img in cases:
    r_np, r_dev = default_rng(77), default_rng(77)
    t0 = time.perf_counter(); want = np.clip(r_np.poisson(img.astype(np.float32)), 0, 255).astype(np.uint8); t_np = time.perf_counter() - t0
    N.np_poisson_u8(img, default_rng(5))
    t0 = time.perf_counter(); got = N.np_poisson_u8(img, r_dev); t_dev = time.perf_counter() - t0
    ok = got is not None and np.array_equal(np.asarray(N.host_array(got)), want) and r_np.bit_generator.state == r_dev.bit_generator.state
    print(name, img.size, 'flags', N.np_poisson_last_flags(), 'numpy %.1f ms device %.1f ms' % (t_np * 1e3, t_dev * 1e3), 'EQUAL' if ok else 'MISMATCH/declined', flush=True)
