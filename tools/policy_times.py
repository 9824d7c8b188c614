#!/usr/bin/env python3
"""Wall time of every policy of the RandomDistortion table on one 1024^2 RGB page (host array in, host array out; levels 1 .. 10,
several seeds): where a page's distortion time can still go.  Usage: tools/policy_times.py [size] > profiles/<tag>_policy_times.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from numpy.random import default_rng

from vkit_amd import _native as N
from vkit_amd.element import Image
from vkit_amd.mechanism.distortion_policy import random_distortion as RD

SIZE = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = default_rng(1)
page = np.full((SIZE, SIZE, 3), 255, np.uint8)
for k in range(0, SIZE - 40, 40):
    page[k + 8:k + 24, 32:SIZE - 32] = g.integers(0, 60, (16, SIZE - 64, 3), dtype=np.uint8) * (g.random((16, SIZE - 64, 1)) < 0.4)
img = Image(mat=page)
factory = RD.RandomDistortionFactory()
out = {}
for fam in (factory.photometric_policy_factories, factory.geometric_policy_factories):
    for pf in fam:
        policy = pf.create()
        times = []
        for seed in range(8):
            level = 1 + seed % 10
            try:
                t0 = time.perf_counter()
                res = policy.distort(level, rng=default_rng(seed), image=img)
                _ = np.asarray(res.image.mat)[0, 0]
                times.append(time.perf_counter() - t0)
            except Exception as exc:       # an out-of-path member configured to raise, etc.
                out[policy.name] = {'error': repr(exc)[:120]}
                break
        if times:
            times = sorted(times[1:])     # the first call warms tables and pools
            out[policy.name] = {'median_ms': round(times[len(times) // 2] * 1e3, 2), 'max_ms': round(times[-1] * 1e3, 2)}
N.default_ctx().sync()
print(json.dumps({'size': SIZE, 'policies': dict(sorted(out.items(), key=lambda kv: -kv[1].get('median_ms', 0)))}))
