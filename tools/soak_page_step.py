#!/usr/bin/env python3
"""PageDistortionStep.run against the oracle replay over many seeds (tests/test_gpu_composite.py::test_page_distortion_step with a longer seed
range): image, active mask, the inactive-region fill and the four label plane sets of every page.  Usage: tools/soak_page_step.py [seeds256] [seeds1024]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_composite as TC
from vkit_amd import _native as N

a = int(sys.argv[1]) if len(sys.argv) > 1 else 150
b = int(sys.argv[2]) if len(sys.argv) > 2 else 20
t0 = time.time()
TC.test_page_distortion_step(N, 256, 24, a)
t1 = time.time()
TC.test_page_distortion_step(N, 1024, 64, b)
print(f'soak_page_step ok: {a} pages of 256^2 ({t1 - t0:.0f} s), {b} pages of 1024^2 ({time.time() - t1:.0f} s), every element equal to the oracle replay')
