#!/usr/bin/env python3
"""PCIe-inclusive rate of the C3 chain: host numpy arrays in, host numpy arrays out, through
the same ChainBatch.  Upload (pageable numpy memory, synchronous vkx_upload) + one pass + download of every result.
This is NOT the bench value (bench.py times with inputs resident in HBM)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time

import numpy as np

sys.argv = [sys.argv[0]]
import bench  # noqa: E402
from vkit_amd import _native  # noqa: E402
from vkit_amd.batch import ChainBatch  # noqa: E402

B, SIZE = 32, 2048
states = [bench.make_state(i, SIZE) for i in range(B)]
images = [np.random.default_rng(1000 + i).integers(0, 256, (SIZE, SIZE, 3), dtype=np.uint8) for i in range(B)]
noises = [bench._noise_plane((5000 + i, tuple(s.result_shape) + (3,))) for i, s in enumerate(states)]
ctx = _native.Context(0)
for attempt in range(2):  # first pass warms the allocator and the kernels
    t0 = time.perf_counter()
    batch = ChainBatch(ctx)
    for img, st, nz in zip(images, states, noises):
        batch.add(img, st, blur_sigma=bench.BLUR_SIGMA, hue_delta=bench.HUE_DELTA, noise=nz)
    t1 = time.perf_counter()
    batch.run()
    ctx.sync()
    t2 = time.perf_counter()
    outs = [batch.result(i) for i in range(B)]
    t3 = time.perf_counter()
    batch.close()
up = sum(a.nbytes for a in images) + sum(a.nbytes for a in noises)
down = sum(a.nbytes for a in outs)
print(json.dumps({'images': B, 'upload_s': t1 - t0, 'run_s': t2 - t1, 'download_s': t3 - t2,
                  'upload_GBps': up / (t1 - t0) / 1e9, 'download_GBps': down / (t3 - t2) / 1e9,
                  'pcie_inclusive_Mpx_s': B * SIZE * SIZE / (t3 - t0) / 1e6}))
