#!/usr/bin/env python3
"""Which host <-> device operations one C4-shaped page issues through the reference's step objects: the Python-level wrappers of
vkit_amd._native.Context counted per page and per step (upload / download / to_device / memset via dev_zeros), with byte sizes.
Usage: tools/probes/page_transfers.py [pages]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from numpy.random import default_rng

from vkit_amd import _native as N
from vkit_amd.pipeline import text_detection as T
from vkit_amd.pipeline.text_detection.synthetic_page import synthetic_page_input

PAGES = int(sys.argv[1]) if len(sys.argv) > 1 else 40
counts = collections.Counter()
bytes_ = collections.Counter()
step = ['?']


def wrap(cls, name, size_of):
    orig = getattr(cls, name)

    def f(self, *a, **k):
        n = size_of(*a, **k)
        bucket = '<=4KB' if n <= 4096 else '<=256KB' if n <= (256 << 10) else '>256KB'
        counts[(step[0], name, bucket)] += 1
        bytes_[(step[0], name, bucket)] += n
        return orig(self, *a, **k)
    setattr(cls, name, f)


wrap(N.Context, 'upload', lambda dptr, array: np.asarray(array).nbytes)
wrap(N.Context, 'download', lambda dptr, array: array.nbytes)
wrap(N.Context, 'upload_async', lambda dptr, array: array.nbytes)
wrap(N.Context, 'download_async', lambda dptr, array: array.nbytes)
wrap(N.Context, 'copy_in', lambda dptr, array, stream=0: array.nbytes)
wrap(N.Context, 'copy_out', lambda dptr, array, stream=0: array.nbytes)
wrap(N.Context, 'sync', lambda: 0)
wrap(N.Context, 'sync_stream', lambda s: 0)
_zeros = N.dev_zeros


def dev_zeros(shape, dtype=np.uint8, ctx=None):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    counts[(step[0], 'dev_zeros', '>256KB' if n > (256 << 10) else '<=256KB')] += 1
    return _zeros(shape, dtype, ctx)


N.dev_zeros = dev_zeros
# every C entry point: count calls per name
calls = collections.Counter()
real_lib = N.lib()


class Counting:
    def __getattr__(self, name):
        fn = getattr(real_lib, name)

        def g(*a):
            calls[(step[0], name)] += 1
            return fn(*a)
        return g


N.lib = lambda: Counting()
step_input = synthetic_page_input(seed=3, size=1024, n_lines=64)
assembler = T.page_assembler_step_factory.create()
distortion = T.page_distortion_step_factory.create()
resizing = T.page_resizing_step_factory.create()


def page(seed):
    rng = default_rng(seed)
    step[0] = 'assembler'
    a = assembler.run(step_input, rng)
    step[0] = 'distortion'
    d = distortion.run(T.PageDistortionStepInput(a), rng)
    step[0] = 'resizing'
    r = resizing.run(T.PageResizingStepInput(d), rng)
    return int(r.page_image.mat[0, 0, 0]) + int(r.page_char_mask.mat.sum() > 0)


for s in range(3):
    page(1000 + s)
counts.clear(); bytes_.clear(); calls.clear()
for s in range(PAGES):
    page(s)
print('per page (mean over %d pages):' % PAGES)
for key in sorted(counts):
    print('  %-11s %-15s %-8s %6.2f calls  %9.0f bytes' % (key[0], key[1], key[2], counts[key] / PAGES, bytes_[key] / PAGES))
print('C entry points per page:')
for key, v in sorted(calls.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print('  %-11s %-40s %6.2f' % (key[0], key[1], v / PAGES))
