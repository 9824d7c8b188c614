"""Which order of operations numpy / OpenBLAS use in the per-vertex part of camera_cubic_curve's state (host probe)."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from numpy.random import default_rng
from vkit_amd.mechanism import distortion as D
from vkit_amd.mechanism.distortion.geometric import camera as C
from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam

f32, f64 = np.float32, np.float64

def fma32(a, b, c):
    return (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)

bad = {'fma_k01': 0, 'fma_k10': 0, 'nofma': 0}
total = 0
for seed in range(300):
    rng = default_rng(seed)
    size = int(rng.integers(200, 2100)); size2 = int(rng.integers(200, 2100))
    level = int(rng.integers(1, 11))
    gen = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), level)
    cfg = gen((size, size2), rng)
    strat = C.CameraCubicCurvePoint2dTo3dStrategy(size, size2, cfg.curve_alpha, cfg.curve_beta, cfg.curve_direction, cfg.curve_scale)
    pts = np.stack([rng.integers(0, size2, 5000), rng.integers(0, size, 5000)], axis=1).astype(f32)
    want = np.matmul(strat.rotation_mat, pts.transpose())[0]
    a0, a1 = strat.rotation_mat[0, 0], strat.rotation_mat[0, 1]
    x, y = pts[:, 0], pts[:, 1]
    c1 = fma32(np.full_like(x, a1), y, (a0 * x).astype(f32))
    c2 = fma32(np.full_like(x, a0), x, (a1 * y).astype(f32))
    c3 = ((a0 * x).astype(f32) + (a1 * y).astype(f32)).astype(f32)
    bad['fma_k01'] += int((c1 != want).sum()); bad['fma_k10'] += int((c2 != want).sum()); bad['nofma'] += int((c3 != want).sum())
    total += 5000
print(total, bad)
# np.cos on a float64 scalar against libm
xs = default_rng(1).uniform(0, np.pi, 200000)
print('np.cos != math.cos:', sum(float(np.cos(v)) != math.cos(v) for v in xs[:50000]), 'np.sin != math.sin:', sum(float(np.sin(v)) != math.sin(v) for v in xs[:50000]))
