#!/usr/bin/env python3
"""Per-kernel roofline table of the single-stage entry points (device resident, 2048^2 RGB unless noted): mean
HIP-event duration over 10 launches, algorithmic bytes, GB/s and fraction of the 8 TB/s peak."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from vkit_amd import _native as N

ctx = N.Context(0)
lib = N.lib()
rng = np.random.default_rng(0)
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048     # 8192: 201 MB per plane, past the 256 MiB Infinity Cache
S = H * W
img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)


def dev(a):
    p = ctx.malloc(a.nbytes)
    ctx.upload(p, np.ascontiguousarray(a))
    return p


d_img, d_out = dev(img), ctx.malloc(img.nbytes * 2)
d_noise = dev(rng.integers(-30, 30, (H, W, 3)).astype(np.int16))
d_f64 = dev(rng.normal(0, 0.2, (H, W, 3)))
d_sel = dev(rng.integers(0, 3, (H, W)).astype(np.uint8))
mx = (np.arange(W, dtype=np.float32)[None, :] + 0.37).repeat(H, 0)
my = (np.arange(H, dtype=np.float32)[:, None] + 0.61).repeat(W, 1)
d_mx, d_my = dev(mx), dev(my)
py, px = np.indices((H, W)).astype(np.int32)
d_py, d_px = dev(np.roll(py, 3, 0)), dev(np.roll(px, 5, 1))
d_hist = ctx.malloc(4 * 256 * 3)
lut = np.stack([np.arange(256, dtype=np.uint8)[::-1]] * 3)
M = (ctypes.c_double * 6)(0.9, -0.1, 30.0, 0.1, 0.9, 10.0)
color = (ctypes.c_uint8 * 4)(10, 20, 30, 0)

cases = [
    ('k_sample_u8 (remap)', lambda: lib.vkx_remap_u8_dev(ctx.handle, d_img, H, W, 3, W * 3, d_mx, d_my, W, d_out, H, W, W * 3), 3 * S + 3 * S + 8 * S),
    ('k_sample_u8 (warpAffine)', lambda: lib.vkx_warp_affine_u8_dev(ctx.handle, d_img, H, W, 3, W * 3, M, d_out, H, W, W * 3), 6 * S),
    ('k_gaussian_blur (5x5)', lambda: lib.vkx_gaussian_blur_u8_dev(ctx.handle, d_img, H, W, 3, W * 3, 5, 1.0, d_out, W * 3), 6 * S),
    ('k_hsv (color_shift)', lambda: lib.vkx_color_shift_rgb_dev(ctx.handle, d_img, H, W, W * 3, 37, d_out, W * 3), 6 * S),
    ('k_cvt (brightness_shift)', lambda: lib.vkx_brightness_shift_rgb_dev(ctx.handle, d_img, H, W, W * 3, 20, d_out, W * 3), 6 * S),
    ('k_mean_shift', lambda: lib.vkx_mean_shift_u8_dev(ctx.handle, d_img, H, W, 3, W * 3, 40, 0, 0, 0, 0, d_out, W * 3), 6 * S),
    ('k_add_noise', lambda: lib.vkx_add_noise_i16_dev(ctx.handle, d_img, H, W, 3, W * 3, d_noise, W * 3, d_out, W * 3), 12 * S),
    ('k_speckle_noise', lambda: lib.vkx_speckle_noise_u8_dev(ctx.handle, d_img, H, W, 3, W * 3, d_f64, W * 3, d_out, W * 3), 30 * S),
    ('k_impulse_noise', lambda: lib.vkx_impulse_noise_u8_dev(ctx.handle, d_img, H, W, 3, W * 3, d_sel, W, d_out, W * 3), 7 * S),
    ('k_pointwise (complement)', lambda: lib.vkx_pointwise_u8_dev(ctx.handle, d_img, H, W, 3, W * 3, 0, -1, 0, 0, d_out, W * 3), 6 * S),
    ('k_line_streak', lambda: lib.vkx_line_streak_u8_dev(ctx.handle, d_out, H, W, 3, W * 3, 2, 20, 0, 0, color, 0.5, 1, 1), 0.19 * 6 * S),
    ('k_histogram', lambda: lib.vkx_histogram_u8_dev(ctx.handle, d_img, H, W, 3, W * 3, d_hist), 3 * S),
    ('k_apply_lut', lambda: lib.vkx_apply_lut_u8_dev(ctx.handle, d_img, H, W, 3, W * 3, lut.ctypes.data, 0, d_out, W * 3), 6 * S),
    ('k_gather', lambda: lib.vkx_gather_u8_dev(ctx.handle, d_img, H, W, 3, W * 3, d_py, d_px, W, d_out, H, W, W * 3), 6 * S + 8 * S),
    ('k_resize_cubic (x1.05)', lambda: lib.vkx_resize_cubic_u8_dev(ctx.handle, d_img, H, W, 3, W * 3, d_out, H * 21 // 20, W * 21 // 20, W * 21 // 20 * 3), 3 * S + 3 * (H * 21 // 20) * (W * 21 // 20)),
    ('k_resize_linear (x0.37)', lambda: lib.vkx_resize_u8_dev(ctx.handle, d_img, H, W, 3, W * 3, d_out, H * 37 // 100, W * 37 // 100, W * 37 // 100 * 3, 1), 3 * S + 3 * (H * 37 // 100) * (W * 37 // 100)),
]
rows = []
for name, fn, nbytes in cases:
    N.check(fn())
    ctx.sync()
    ctx.set_timing(True)
    ctx.reset_timings()
    for _ in range(10):
        N.check(fn())
    t = ctx.timings()
    ctx.set_timing(False)
    key = max(t, key=lambda k: t[k][0])
    ms = t[key][0] / t[key][1]
    rows.append({'kernel': name, 'timed': key, 'ms': round(ms, 4), 'algorithmic_MB': round(nbytes / 1e6, 1),
                 'GBps': round(nbytes / ms / 1e6), 'frac_of_8TBps': round(nbytes / ms / 1e6 / 8000, 3)})
print(json.dumps(rows))
