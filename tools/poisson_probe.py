#!/usr/bin/env python3
"""rng.poisson on the device (vkx_np_poisson_u8) against numpy itself: values, stream position, time.  Usage: tools/poisson_probe.py [size]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from numpy.random import default_rng

from vkit_amd import _native as N

SIZE = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = N.Context(int(os.environ['VKX_DEVICE'])) if 'VKX_DEVICE' in os.environ else N.default_ctx()


def cases():
    g = default_rng(7)
    yield 'uniform bytes', g.integers(0, 256, (SIZE, SIZE, 3), dtype=np.uint8)
    page = np.full((SIZE, SIZE, 3), 255, np.uint8)
    for k in range(0, SIZE - 40, 40):
        page[k + 8:k + 24, 32:SIZE - 32] = g.integers(0, 60, (16, SIZE - 64, 3), dtype=np.uint8) * (g.random((16, SIZE - 64, 1)) < 0.4)
    yield 'page', page
    yield 'all 255', np.full((SIZE, SIZE, 3), 255, np.uint8)
    yield 'all 9', np.full((SIZE // 2, SIZE // 2, 3), 9, np.uint8)
    yield 'all 10', np.full((SIZE // 2, SIZE // 2, 3), 10, np.uint8)
    yield 'zeros', np.zeros((64, 64, 3), np.uint8)
    yield 'dark 0..12', g.integers(0, 13, (SIZE // 2, SIZE // 2, 3), dtype=np.uint8)
    yield 'gray plane', g.integers(0, 256, (SIZE // 2, SIZE // 2 + 3), dtype=np.uint8)
    for n in (1, 2, 31, 32, 33, 1000):
        yield f'n={n}', g.integers(0, 256, (n,), dtype=np.uint8)


rows = []
for name, img in cases():
    r_np, r_dev = default_rng(11), default_rng(11)
    r_np.random(3); r_dev.random(3)
    t0 = time.perf_counter()
    want = np.clip(r_np.poisson(img.astype(np.float32)), 0, 255).astype(np.uint8)
    t_np = time.perf_counter() - t0
    N.np_poisson_u8(img, default_rng(5), ctx=ctx)      # warm-up: tables, scratch
    ctx.set_timing(True); ctx.reset_timings()
    t0 = time.perf_counter()
    got = N.np_poisson_u8(img, r_dev, ctx=ctx)
    t_dev = time.perf_counter() - t0
    k = ctx.timings(); ctx.set_timing(False)
    row = {'case': name, 'n': int(img.size), 'flags': N.np_poisson_last_flags(), 'numpy_ms': round(t_np * 1e3, 2), 'device_ms': round(t_dev * 1e3, 2)}
    if got is not None:
        got = np.asarray(N.host_array(got))
        row['equal'] = bool(np.array_equal(got, want))
        row['mismatches'] = int((got != want).sum())
        row['stream_equal'] = bool(r_np.bit_generator.state == r_dev.bit_generator.state and r_np.random() == r_dev.random())
        row['kernels_ms'] = {n: [round(v[0], 3), v[1]] for n, v in k.items() if n.startswith('k_pz')}
    rows.append(row)
    print(json.dumps(row), flush=True)
ok = all(r.get('equal') and r.get('stream_equal') for r in rows)
print('ALL EQUAL' if ok else 'MISMATCH')
# last line: the summary bench.py's other_configs takes
print(json.dumps({'rng_poisson_on_device': {r['case']: {'n': r['n'], 'numpy_ms': r['numpy_ms'], 'device_ms': r['device_ms'], 'equal_to_numpy': bool(r.get('equal') and r.get('stream_equal'))}
                                            for r in rows if not r['case'].startswith('n=')}, 'all_equal': ok}))
