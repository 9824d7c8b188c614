#!/usr/bin/env python3
"""Device dispatches of one C4-shaped page, attributed to the C entry point that issued them: every call through the C ABI runs inside
a ROCTx range named after the entry point and the step (assembler / distortion / resizing); under
    rocprofv3 --kernel-trace --marker-trace --kernel-rename
every dispatch -- the runtime's own copy / fill kernels included -- then carries that name (tools/probes/page_dispatches.sh counts them).
Usage: page_dispatches.py [pages]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from numpy.random import default_rng

from vkit_amd import _native as N
from vkit_amd.pipeline import text_detection as T
from vkit_amd.pipeline.text_detection.synthetic_page import synthetic_page_input

PAGES = int(sys.argv[1]) if len(sys.argv) > 1 else 40
roctx = ctypes.CDLL('librocprofiler-sdk-roctx.so')
roctx.roctxRangePushA.argtypes = [ctypes.c_char_p]
step = ['warmup']
real_lib = N.lib()
cache = {}


class Ranged:
    def __getattr__(self, name):
        fn = getattr(real_lib, name)

        def g(*a):
            key = (step[0], name)
            label = cache.get(key)
            if label is None:
                label = cache[key] = f'{step[0]}:{name}'.encode()
            roctx.roctxRangePushA(label)
            try:
                return fn(*a)
            finally:
                roctx.roctxRangePop()
        return g


N.lib = lambda: Ranged()
step_input = synthetic_page_input(seed=3, size=1024, n_lines=64)
assembler = T.page_assembler_step_factory.create()
distortion = T.page_distortion_step_factory.create()
resizing = T.page_resizing_step_factory.create()


def page(seed, tag):
    rng = default_rng(seed)
    step[0] = tag + 'assembler'
    a = assembler.run(step_input, rng)
    step[0] = tag + 'distortion'
    d = distortion.run(T.PageDistortionStepInput(a), rng)
    step[0] = tag + 'resizing'
    r = resizing.run(T.PageResizingStepInput(d), rng)
    return int(r.page_image.mat[0, 0, 0]) + int(r.page_char_mask.mat.sum() > 0)


for s in range(3):
    page(1000 + s, 'warmup_')
for s in range(PAGES):
    page(s, '')
print('pages', PAGES)
