#!/usr/bin/env python3
"""k_project_points alone: kernel time by HIP events for n points on (a) an identity lattice, (b) a policy camera_cubic_curve lattice,
(c) the same with ONE degenerate cell (three collinear corners) that holds one point -- how much of its 70 us per page is the eight-lane
Jacobi solve of degenerate cells and how much the closed form."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from numpy.random import default_rng

from vkit_amd import _native as N
from vkit_amd.mechanism import distortion as D
from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam

ctx = N.default_ctx()
size, grid = 1024, 15
rng = default_rng(0)
st = D.camera_cubic_curve.generate_state(P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), 5)((size, size), default_rng(3)), (size, size))
sv = np.asarray(st.src_image_grid.vertices, np.int32)
dv = np.asarray(st.dst_image_grid.vertices, np.int32)


def run(name, sv, dv, n):
    pts = rng.uniform(1, size - 2, (n, 2))
    pi = np.rint(pts).astype(np.int32)
    for _ in range(3):
        N.project_points(sv, dv, grid, pi, pts)
    ctx.set_timing(True); ctx.reset_timings()
    for _ in range(20):
        N.project_points(sv, dv, grid, pi, pts)
    t = ctx.timings(); ctx.set_timing(False)
    print(f'{name:34s} n={n:5d}: k_project_points {t["k_project_points"][0] / 20 * 1e3:7.1f} us')


for n in (64, 900, 2700):
    run('identity lattice', sv, sv, n)
    run('camera_cubic_curve level 5', sv, dv, n)
bad = dv.copy()
bad[1, 1] = (bad[1, 0] + bad[1, 2]) // 2          # three vertices of a row collinear: cells (0,0),(0,1),(1,0),(1,1) touch it
if (bad[1, 0][1] == bad[1, 2][1]):
    bad[1, 1][1] = bad[1, 0][1]
run('... with a collinear vertex', sv, bad, 2700)
