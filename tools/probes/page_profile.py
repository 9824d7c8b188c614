#!/usr/bin/env python3
"""cProfile of whole C4-shaped pages through the reference's step objects in one process (assembler -> distortion -> resizing, resized outputs
read on the host): where the host time of a page goes.  Usage: tools/probes/page_profile.py [pages] [sort]"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from numpy.random import default_rng

from vkit_amd import _native as N
from vkit_amd.pipeline.text_detection.synthetic_page import synthetic_page_input as _synthetic_page_input
from vkit_amd.pipeline import text_detection as T

PAGES = int(sys.argv[1]) if len(sys.argv) > 1 else 48
SORT = sys.argv[2] if len(sys.argv) > 2 else 'tottime'
ctx = N.default_ctx()
step_input = _synthetic_page_input(seed=3, size=1024, n_lines=64)
assembler = T.page_assembler_step_factory.create()
distortion = T.page_distortion_step_factory.create()
resizing = T.page_resizing_step_factory.create()


def page(seed):
    rng = default_rng(seed)
    a = assembler.run(step_input, rng)
    d = distortion.run(T.PageDistortionStepInput(a), rng)
    r = resizing.run(T.PageResizingStepInput(d), rng)
    return int(r.page_image.mat[0, 0, 0]) + int(r.page_char_mask.mat.sum() > 0)


for s in range(4):
    page(1000 + s)
times = []
for s in range(PAGES):
    t0 = time.perf_counter(); page(s); times.append(time.perf_counter() - t0)
print('pages %d: mean %.3f ms median %.3f ms max %.3f ms' % (PAGES, 1e3 * np.mean(times), 1e3 * np.median(times), 1e3 * np.max(times)))
steps = {'assembler': 0.0, 'distortion': 0.0, 'resizing': 0.0}
for s in range(PAGES):
    rng = default_rng(s)
    t0 = time.perf_counter(); a = assembler.run(step_input, rng); t1 = time.perf_counter()
    d = distortion.run(T.PageDistortionStepInput(a), rng); t2 = time.perf_counter()
    r = resizing.run(T.PageResizingStepInput(d), rng); _ = int(r.page_image.mat[0, 0, 0]); t3 = time.perf_counter()
    steps['assembler'] += t1 - t0; steps['distortion'] += t2 - t1; steps['resizing'] += t3 - t2
print({k: round(1e3 * v / PAGES, 3) for k, v in steps.items()})
pr = cProfile.Profile()
pr.enable()
for s in range(PAGES):
    page(s)
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats(SORT).print_stats(60)
print(out.getvalue())
