#!/usr/bin/env python3
"""Second soak: randomised parity of the single-stage entry points against the oracle (tools/soak.py covers the chain,
lattice remaps, resizes, polygons, composite, filter2D, point projection).  Usage: tools/soak2.py <seconds> <seed>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from numpy.random import default_rng

import oracle as O
from vkit_amd import _native as N
import vkit_amd.mechanism.distortion as D
from vkit_amd.element import Image

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
counts = {}


def same(got, want, what):
    ok = ((got == want) | ((got != got) & (want != want))).all() if got.dtype.kind == 'f' else (got == want).all()
    assert got.shape == want.shape and ok, what
    counts[what[0]] = counts.get(what[0], 0) + 1


def image(h, w, cn):
    return rng.integers(0, 256, (h, w) if cn == 1 else (h, w, cn), dtype=np.uint8)


while time.time() - t0 < budget:
    h, w = int(rng.integers(1, 260)), int(rng.integers(1, 260))
    cn = int(rng.choice([1, 3, 4]))
    src = image(h, w, cn)
    rgb = image(h, w, 3)
    plane = (rng.random((h, w), dtype=np.float32) * 50).astype(np.float32)
    dh, dw = int(rng.integers(1, 260)), int(rng.integers(1, 260))
    # remap with wild maps
    mx = rng.uniform(-8, w + 8, (dh, dw)).astype(np.float32)
    my = rng.uniform(-8, h + 8, (dh, dw)).astype(np.float32)
    wild = rng.random((dh, dw))
    mx[wild < 0.01] = np.nan
    my[(wild > 0.01) & (wild < 0.02)] = np.inf
    mx[(wild > 0.02) & (wild < 0.03)] = -1e30
    my[(wild > 0.03) & (wild < 0.04)] = 3e9
    same(N.remap(src, mx, my), O.remap(src, mx, my), ('remap', src.shape, (dh, dw)))
    same(N.remap(plane, mx, my), O.remap(plane, mx, my), ('remap_f32', (h, w), (dh, dw)))
    # warps
    ang = rng.uniform(0, 6.28)
    sc = rng.uniform(0.3, 2.5)
    M = np.array([[sc * np.cos(ang), -sc * np.sin(ang), rng.uniform(-30, 60)],
                  [sc * np.sin(ang), sc * np.cos(ang), rng.uniform(-30, 60)]], np.float32).astype(np.float64)
    same(N.warp_affine(src, M, (dw, dh)), O.warp_affine(src, M, (dw, dh)), ('warp_affine', src.shape, (dh, dw)))
    same(N.warp_affine(plane, M, (dw, dh)), O.warp_affine(plane, M, (dw, dh)), ('warp_affine_f32', (h, w)))
    P = np.vstack([M, [rng.uniform(-2e-3, 2e-3), rng.uniform(-2e-3, 2e-3), 1.0]])
    same(N.warp_perspective(src, P, (dw, dh)), O.warp_perspective(src, P, (dw, dh)), ('warp_perspective', src.shape))
    same(N.warp_perspective(plane, P, (dw, dh)), O.warp_perspective(plane, P, (dw, dh)), ('warp_perspective_f32', (h, w)))
    # photometric single stages
    if min(h, w) > 1:
        k = int(rng.choice([3, 5, 7]))
        sigma = float(rng.uniform(0.3, 2.5))
        same(N.gaussian_blur(src, k, sigma), O.gaussian_blur(src, k, sigma), ('gaussian_blur', src.shape, k, sigma))
    delta = int(rng.integers(-300, 300))
    same(N.color_shift_rgb(rgb, delta), O.color_shift_rgb(rgb, delta), ('color_shift', delta))
    thr = None if rng.random() < 0.5 else int(rng.integers(0, 256))
    chans = None if rng.random() < 0.5 or cn == 1 else sorted(set(int(c) for c in rng.integers(0, cn, 2)))
    cyc = bool(rng.random() < 0.5)
    d8 = int(rng.integers(-255, 256))
    same(N.mean_shift(src, d8, threshold=thr, channels=chans, cycle=cyc), O.mean_shift(src, d8, thr, chans, cyc),
         ('mean_shift', d8, thr, chans, cyc))
    noise = rng.integers(-32768, 32768, src.shape).astype(np.int16) if rng.random() < 0.2 else \
        np.round(rng.normal(0, 40, src.shape)).astype(np.int16)
    same(N.add_noise_i16(src, noise), O.add_noise_i16(src, noise), ('add_noise',))
    t, g = int(rng.integers(1, 6)), int(rng.integers(0, 9))
    dt, dg = (int(rng.integers(1, 5)), int(rng.integers(1, 5))) if rng.random() < 0.5 else (0, 0)
    col = tuple(int(v) for v in rng.integers(0, 256, cn))
    alpha = float(rng.choice([1.0, rng.random()]))
    ev, eh = bool(rng.random() < 0.8), bool(rng.random() < 0.8)
    same(N.line_streak(src, t, g, dt, dg, col, alpha, ev, eh),
         O.line_streak(src, t, g, dt, dg, col, alpha, ev, eh), ('line_streak', t, g, dt, dg, alpha, ev, eh))
    cthr = None if rng.random() < 0.5 else int(rng.integers(0, 256))
    lte = bool(rng.random() < 0.5)
    same(N.pointwise(src, N.POINT_COMPLEMENT, -1 if cthr is None else cthr, int(lte), channels=chans),
         O.complement(src, cthr, lte, chans), ('complement', cthr, lte, chans))
    bits = int(rng.integers(0, 8))
    same(N.pointwise(src, N.POINT_POSTERIZE, bits, channels=chans), O.posterization(src, bits, chans), ('posterize', bits))
    if cn > 1:
        perm = [int(v) for v in rng.integers(0, cn, cn)]
        same(N.permute_channels(src, perm), O.permute_channels(src, perm), ('permute', perm))
    sel = rng.integers(0, 3, (h, w), dtype=np.uint8)
    same(N.impulse_noise(src, sel), O.impulse_noise(src, sel), ('impulse',))
    sp = rng.normal(0, float(rng.uniform(0.1, 2.0)), src.shape)
    same(N.speckle_noise(src, sp), O.speckle_noise(src, sp), ('speckle',))
    for code, fn in ((N.CVT_RGB2HSV_FULL, O.rgb2hsv_full), (N.CVT_HSV2RGB_FULL, O.hsv2rgb_full),
                     (N.CVT_RGB2HLS_FULL, O.rgb2hls_full), (N.CVT_HLS2RGB_FULL, O.hls2rgb_full), (N.CVT_RGB2GRAY, O.rgb2gray)):
        same(N.cvt_color(rgb, code), fn(rgb), ('cvt', code))
    bd = int(rng.integers(-255, 256))
    same(N.brightness_shift_rgb(rgb, bd), O.brightness_shift_rgb(rgb, bd), ('brightness', bd))
    ratio = float(rng.random())
    same(N.color_balance_rgb(rgb, ratio), O.color_balance_rgb(rgb, ratio), ('color_balance', ratio))
    lut = rng.integers(0, 256, (cn, 256), dtype=np.uint8)
    want = src.copy().reshape(h, w, cn)
    for c in range(cn):
        want[:, :, c] = lut[c][want[:, :, c]]
    same(N.apply_lut(src, lut), want.reshape(src.shape), ('apply_lut',))
    py = rng.integers(0, h, (dh, dw)).astype(np.int32)
    px = rng.integers(0, w, (dh, dw)).astype(np.int32)
    same(N.gather(src, py, px), src[py, px], ('gather',))
    big = rng.integers(-2 ** 40, 2 ** 40, src.shape) if rng.random() < 0.3 else rng.poisson(rng.uniform(0, 300), src.shape)
    same(N.saturate_i64(big), np.clip(big, 0, 255).astype(np.uint8), ('saturate_i64',))
    hist = N.histogram(src).reshape(cn, 256)
    ref = np.stack([np.bincount(src.reshape(h, w, cn)[:, :, c].ravel(), minlength=256) for c in range(cn)])
    same(hist.astype(np.int64), ref.astype(np.int64), ('histogram',))
    # the two filter2D operators end to end
    radius = int(rng.integers(1, 4))
    aa = float(rng.uniform(0.4, 1.2))
    got = D.defocus_blur.distort(D.DefocusBlurConfig(radius=radius, anti_aliasing_sigma=aa), image=Image(mat=rgb)).image.mat
    same(got, O.defocus_blur(rgb, radius, aa), ('defocus', radius, aa))
    angle = int(rng.integers(0, 360))
    got = D.motion_blur.distort(D.MotionBlurConfig(radius=radius, angle=angle, anti_aliasing_sigma=aa), image=Image(mat=rgb)).image.mat
    same(got, O.motion_blur(rgb, radius, angle, aa), ('motion', radius, angle, aa))
print('soak2 ok', counts, round(time.time() - t0), 's')
