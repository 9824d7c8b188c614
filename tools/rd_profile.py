#!/usr/bin/env python3
"""Where RandomDistortion.distort spends its time on 1024^2 pages: wall time per policy name (device work + the host
arithmetic the reference's rng contract leaves on the host: numpy draws for noise / fog / glass shuffles).
Usage: tools/rd_profile.py [out.json]"""
import json, os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from numpy.random import default_rng
from vkit_amd.element import Image
from vkit_amd.mechanism.distortion_policy import random_distortion_factory
from vkit_amd.mechanism.distortion_policy import type as policy_type

P = 1024
pages = [Image(mat=default_rng(100 + i).integers(0, 256, (P, P, 3), dtype=np.uint8)) for i in range(8)]
acc = collections.defaultdict(lambda: [0.0, 0])
orig = policy_type.DistortionPolicy.distort
def timed(self, *a, **k):
    t = time.perf_counter()
    try:
        return orig(self, *a, **k)
    finally:
        rec = acc[self.name]; rec[0] += time.perf_counter() - t; rec[1] += 1
policy_type.DistortionPolicy.distort = timed
rd = random_distortion_factory.create()
rd.distort(default_rng(0), image=pages[0])
acc.clear()
n = 200
t0 = time.perf_counter()
for k in range(n):
    rd.distort(default_rng(k), image=pages[k % len(pages)])
total = time.perf_counter() - t0
rows = sorted(((name, v[0] / v[1] * 1e3, v[1], v[0] / total) for name, v in acc.items()), key=lambda r: -r[3])
out = {'pages': n, 'ms_per_page': total / n * 1e3, 'policies': [{'name': r[0], 'ms_per_call': round(r[1], 3), 'calls': r[2], 'share': round(r[3], 3)} for r in rows]}
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)
