#!/usr/bin/env python3
"""Soak of the round-4 device members against their host restatements: the fog density plane (np_fog_mask vs
generate_diamond_square_mask + stretch), glass_blur's shuffle planes (glass_shuffle_planes_dev vs glass_shuffle_planes) -- planes and
generator position --, for random shapes and parameters.  Usage: tools/soak9.py <seconds> <seed> > profiles/<tag>_soak9.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from numpy.random import default_rng

from vkit_amd import _native as N
from vkit_amd.mechanism.distortion.photometric.blur import glass_shuffle_planes
from vkit_amd.mechanism.distortion.photometric.effect import generate_diamond_square_mask

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
g = default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
counts = {'fog': 0, 'glass': 0}
while time.time() - t0 < budget:
    shape = (int(g.integers(1, 1500)), int(g.integers(1, 1500)))
    seed, skip = int(g.integers(1 << 30)), int(g.integers(0, 50))
    # fog
    rough, lo, hi = float(g.random()), float(g.random() * 0.4), float(0.5 + g.random() * 0.5)
    r_np, r_dev = default_rng(seed), default_rng(seed)
    r_np.random(skip); r_dev.random(skip)
    got = N.np_fog_mask(shape, rough, lo, hi, r_dev)
    if got is not None:
        want = np.array(generate_diamond_square_mask(shape, rough, r_np), dtype=np.float32)
        want -= want.min(); want /= want.max(); want *= (hi - lo); want += lo
        assert np.array_equal(np.asarray(N.host_array(got)).view(np.uint32), want.view(np.uint32)), ('fog', shape, rough, seed)
        assert r_np.bit_generator.state == r_dev.bit_generator.state, ('fog stream', shape, seed)
        counts['fog'] += 1
    # glass
    delta, loop = int(g.integers(1, 9)), int(g.integers(1, 8))
    r_np, r_dev = default_rng(seed + 1), default_rng(seed + 1)
    wy, wx = glass_shuffle_planes(shape, delta, loop, r_np)
    gy, gx = N.glass_shuffle_planes_dev(shape, delta, loop, r_dev)
    assert np.array_equal(np.asarray(N.host_array(gy)), wy) and np.array_equal(np.asarray(N.host_array(gx)), wx), ('glass', shape, delta, loop, seed)
    assert r_np.bit_generator.state == r_dev.bit_generator.state, ('glass stream', shape, seed)
    counts['glass'] += 1
print(json.dumps({'seconds': round(time.time() - t0, 1), 'equal_to_host_restatement': counts}))
