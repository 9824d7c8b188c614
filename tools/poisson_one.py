#!/usr/bin/env python3
"""One device rng.poisson call on a 1024^2 RGB image of uniform random bytes (32 distinct rates per block: the table pass at its
heaviest), for profilers.  Usage: tools/poisson_one.py [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from numpy.random import default_rng

from vkit_amd import _native as N

img = default_rng(7).integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    out = N.np_poisson_u8(img, default_rng(11 + k))
    assert out is not None
print('ok')
