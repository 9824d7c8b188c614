#!/usr/bin/env python3
"""How many device dispatches one host <-> device operation of the operator API costs on this stack (run under rocprofv3 --kernel-trace
--memory-copy-trace, one op kind per process: tools/probes/copy_kinds.sh).  Usage: copy_kinds.py <op> [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from vkit_amd import _native as N

op = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = N.default_ctx()
lib = N.lib()
MB3 = 1024 * 1024 * 3
pageable = np.random.default_rng(0).integers(0, 256, MB3, dtype=np.uint8)
pinned = ctx.pinned_empty((MB3,), np.uint8)
pinned[:] = pageable
small = np.arange(256, dtype=np.uint8)
d = ctx.malloc(MB3)
d2 = ctx.malloc(MB3)
out_pageable = np.empty(MB3, np.uint8)
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    if op == 'upload_3mb_pageable':
        ctx.upload(d, pageable)
    elif op == 'upload_3mb_pinned':
        ctx.upload(d, pinned)
    elif op == 'upload_256b_pageable':
        ctx.upload(d, small)
    elif op == 'upload_64kb_pageable':
        ctx.upload(d, pageable[:65536])
    elif op == 'upload_1mb_pageable':
        ctx.upload(d, pageable[:1 << 20])
    elif op == 'download_3mb_pageable':
        ctx.download(d, out_pageable)
    elif op == 'download_3mb_pinned':
        ctx.download(d, pinned)
    elif op == 'download_4b_pageable':
        ctx.download(d, out_pageable[:4])
    elif op == 'memset_1mb':
        N.check(lib.vkx_memset(ctx.handle, N.c_void_p(d), 0, 1 << 20))
        ctx.sync()
    elif op == 'memset_4b':
        N.check(lib.vkx_memset(ctx.handle, N.c_void_p(d), 0, 4))
        ctx.sync()
    elif op == 'd2d_3mb':
        N.check(lib.vkx_memcpy_async(ctx.handle, int(N.STREAM_COMPUTE), N.c_void_p(d2), N.c_void_p(d), MB3, 2))
        ctx.sync()
    elif op == 'to_device_3mb':
        a = ctx.to_device(pageable.reshape(1024, 1024, 3))
        del a
    elif op == 'none':
        ctx.sync()
    else:
        raise SystemExit('unknown op ' + op)
ctx.sync()
print(op, 'host us per rep', round((time.perf_counter() - t0) / reps * 1e6, 1))
