#!/usr/bin/env python3
"""Probe: device->host copy rate into pageable vs pinned host memory (decides whether results should be handed out in
pinned arrays)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from vkit_amd import _native as N

ctx = N.Context(0)
hip = ctypes.CDLL(N.HIP_RUNTIME if N.HIP_RUNTIME != 'system' else 'libamdhip64.so')
n = 14 * 1024 * 1024
d = ctx.malloc(n)
ctx.upload(d, np.zeros(n, np.uint8))
for kind in ('pageable', 'pinned', 'pageable_fresh'):
    if kind == 'pinned':
        p = ctypes.c_void_p()
        t0 = time.perf_counter()
        assert hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(n), 0) == 0
        t_alloc = time.perf_counter() - t0
        arr = np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(p.value))
    else:
        t_alloc = 0.0
        arr = np.empty(n, np.uint8)
    times = []
    for rep in range(6):
        if kind == 'pageable_fresh':
            arr = np.empty(n, np.uint8)      # first touch inside the copy, like a result array
        t0 = time.perf_counter()
        ctx.download(d, arr)
        times.append(time.perf_counter() - t0)
    print(kind, 'alloc ms', round(t_alloc * 1e3, 3), 'copy GB/s', [round(n / t / 1e9, 1) for t in times])
