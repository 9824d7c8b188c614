#!/usr/bin/env python3
"""Soak of the device rng.poisson (vkx_np_poisson_u8) against numpy: images of many sizes and value distributions -- flat, pages
with dark text, gradients, photographs' worth of noise, long zero / low-rate stretches --, each from its own stream position.
Every image: values and the generator's state afterwards must be numpy's; declined images (flags != 0) are counted, not errors.
Usage: tools/soak8.py [images] > profiles/<tag>_soak8.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from numpy.random import default_rng

from vkit_amd import _native as N

IMAGES = int(sys.argv[1]) if len(sys.argv) > 1 else 120


def image(g, k):
    h, w = int(g.integers(1, 1400)), int(g.integers(1, 1400))
    cn = (1, 3, 3, 4)[k % 4]
    shape = (h, w) if cn == 1 else (h, w, cn)
    kind = k % 7
    if kind == 0:
        return g.integers(0, 256, shape, dtype=np.uint8)
    if kind == 1:                                              # page: white, dark text rows
        img = np.full(shape, int(g.integers(200, 256)), np.uint8)
        for y in range(0, h - 12, 30):
            img[y + 4:y + 12] = (g.integers(0, 80, img[y + 4:y + 12].shape) * (g.random(img[y + 4:y + 12].shape) < 0.4)).astype(np.uint8)
        return img
    if kind == 2:                                              # gradient across every rate, both regimes in every row
        return np.broadcast_to((np.arange(w) * 255 // max(w - 1, 1)).astype(np.uint8).reshape((1, w) + (1,) * (len(shape) - 2)), shape).copy()
    if kind == 3:                                              # low rates: the multiplication method and zeros
        return g.integers(0, 12, shape, dtype=np.uint8)
    if kind == 4:                                              # a noisy flat field around the PTRS threshold
        return np.clip(g.normal(10, 3, shape), 0, 255).astype(np.uint8)
    if kind == 5:                                              # long constant stretches with steps
        flat = np.repeat(g.integers(0, 256, 1 + h * w * cn // 5000, dtype=np.uint8), 5000)[:h * w * cn]
        return flat.reshape(shape)
    return np.clip(g.normal(128, 60, shape), 0, 255).astype(np.uint8)


g = default_rng(2024)
t0 = time.time()
elements = declined = 0
flags_seen = {}
for k in range(IMAGES):
    img = image(g, k)
    seed, skip = int(g.integers(1 << 30)), int(g.integers(0, 1000))
    r_np, r_dev = default_rng(seed), default_rng(seed)
    r_np.random(skip); r_dev.random(skip)
    got = N.np_poisson_u8(img, r_dev)
    if got is None:
        declined += 1
        flags_seen[str(N.np_poisson_last_flags())] = flags_seen.get(str(N.np_poisson_last_flags()), 0) + 1
        continue
    want = np.clip(r_np.poisson(img.astype(np.float32)), 0, 255).astype(np.uint8)
    assert np.array_equal(np.asarray(N.host_array(got)), want), f'image {k} {img.shape}: values differ'
    assert r_np.bit_generator.state == r_dev.bit_generator.state, f'image {k} {img.shape}: stream position differs'
    elements += img.size
print(json.dumps({'images': IMAGES, 'declined': declined, 'declined_flags': flags_seen, 'elements_equal_to_numpy': int(elements),
                  'seconds': round(time.time() - t0, 1)}))
