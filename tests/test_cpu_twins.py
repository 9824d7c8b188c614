"""CPU twins of the product's entry points (oracle/vkx_cpu.c): `vkx_cpu_X` takes exactly the arguments of `vkx_X`
(include/vkx.h) -- a binding written for libvkx.so reaches the reference's CPU arithmetic by changing the library and the prefix.
Here: (not gpu) every twin exists, takes the product's ctypes signature and gives the oracle's result; (gpu) the same ctypes call on
the product and on the twin, argument for argument, gives the same bytes."""
import ctypes

import numpy as np
import pytest
from numpy.random import default_rng

import oracle as O
from vkit_amd import _native as N

TWINS = ['vkx_remap_u8', 'vkx_remap_f32', 'vkx_warp_affine_u8', 'vkx_grid_to_map', 'vkx_grid_remap', 'vkx_gaussian_blur_u8',
         'vkx_color_shift_rgb', 'vkx_add_noise_i16', 'vkx_line_streak_u8', 'vkx_fill_u8']


def _twin(name):
    fn = getattr(O.lib(), name.replace('vkx_', 'vkx_cpu_', 1))
    fn.argtypes, fn.restype = N._SIGNATURES[name], ctypes.c_int
    return fn


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _lattice(h, w, seed):
    rng = default_rng(seed)
    ys = list(range(0, h, 17)) + [h - 1]
    xs = list(range(0, w, 19)) + [w - 1]
    sv = np.array([[(x, y) for x in xs] for y in ys], np.int32)
    dv = sv + rng.integers(-5, 6, sv.shape).astype(np.int32)
    dv[..., 0] -= dv[..., 0].min()
    dv[..., 1] -= dv[..., 1].min()
    return sv, dv, (int(dv[..., 1].max()) + 1, int(dv[..., 0].max()) + 1)


def _calls(seed=0):
    """(name, argument tuple builder, output arrays) for every twin: the arguments are what a caller of libvkx.so passes."""
    rng = default_rng(seed)
    h, w = 97, 131
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    score = rng.random((h, w), dtype=np.float32)
    sv, dv, (dh, dw) = _lattice(h, w, seed + 1)
    mx, my = O.grid_to_map(sv, dv, (dh, dw))
    noise = np.round(default_rng(seed + 2).normal(0, 11, (h, w, 3))).astype(np.int16)
    M = np.array([0.9, 0.2, 3.0, -0.15, 1.05, 5.0], np.float64)
    out = {}

    def remap_u8(fn, ctx):
        dst = np.zeros((dh, dw, 3), np.uint8)
        rc = fn(ctx, _ptr(img), h, w, 3, w * 3, _ptr(mx), _ptr(my), dw, _ptr(dst), dh, dw, dw * 3)
        return rc, [dst]

    def remap_f32(fn, ctx):
        dst = np.zeros((dh, dw), np.float32)
        rc = fn(ctx, _ptr(score), h, w, w, _ptr(mx), _ptr(my), dw, _ptr(dst), dh, dw, dw)
        return rc, [dst]

    def warp(fn, ctx):
        dst = np.zeros((h + 9, w + 5, 3), np.uint8)
        rc = fn(ctx, _ptr(img), h, w, 3, w * 3, _ptr(M), _ptr(dst), h + 9, w + 5, (w + 5) * 3)
        return rc, [dst]

    def grid_to_map(fn, ctx):
        gx, gy = np.zeros((dh, dw), np.float32), np.zeros((dh, dw), np.float32)
        rc = fn(ctx, _ptr(sv), _ptr(dv), sv.shape[0], sv.shape[1], dh, dw, _ptr(gx), _ptr(gy), dw, None)
        return rc, [gx.view(np.uint32), gy.view(np.uint32)]

    def grid_remap(fn, ctx):
        d_img, d_score = np.zeros((dh, dw, 3), np.uint8), np.zeros((dh, dw), np.float32)
        elems = (N.VkxElem * 2)()
        elems[0].src, elems[0].dst, elems[0].src_stride, elems[0].dst_stride, elems[0].cn, elems[0].is_f32 = img.ctypes.data, d_img.ctypes.data, w * 3, dw * 3, 3, 0
        elems[1].src, elems[1].dst, elems[1].src_stride, elems[1].dst_stride, elems[1].cn, elems[1].is_f32 = score.ctypes.data, d_score.ctypes.data, w, dw, 1, 1
        rc = fn(ctx, elems, 2, h, w, _ptr(sv), _ptr(dv), sv.shape[0], sv.shape[1], dh, dw)
        return rc, [d_img, d_score.view(np.uint32)]

    def blur(fn, ctx):
        dst = np.zeros_like(img)
        rc = fn(ctx, _ptr(img), h, w, 3, w * 3, 5, 1.0, _ptr(dst), w * 3)
        return rc, [dst]

    def hue(fn, ctx):
        dst = np.zeros_like(img)
        rc = fn(ctx, _ptr(img), h, w, w * 3, 37, _ptr(dst), w * 3)
        return rc, [dst]

    def add_noise(fn, ctx):
        dst = np.zeros_like(img)
        rc = fn(ctx, _ptr(img), h, w, 3, w * 3, _ptr(noise), w * 3, _ptr(dst), w * 3)
        return rc, [dst]

    def streak(fn, ctx):
        page = img.copy()
        color = (ctypes.c_uint8 * 4)(10, 200, 30, 0)
        rc = fn(ctx, _ptr(page), h, w, 3, w * 3, 2, 9, 3, 5, ctypes.cast(color, ctypes.c_void_p), 0.6, 1, 1)
        return rc, [page]

    def fill(fn, ctx):
        page = img.copy()
        alpha = (default_rng(seed + 3).random((40, 60), dtype=np.float32) * (default_rng(seed + 4).random((40, 60)) < 0.5)).astype(np.float32)
        value = default_rng(seed + 5).integers(0, 256, (30, 50, 3), dtype=np.uint8)
        layers = (N.VkxLayer * 3)()
        keep = []
        for k, (layer, _keep) in enumerate([N.make_layer((5, 7, 40, 60), 3, (10, 20, 30), alpha=alpha),
                                            N.make_layer((20, 30, 30, 50), 3, value, alpha=0.4),
                                            N.make_layer((0, 0, h, w), 3, (90, 90, 90), mode=N.FILL_KEEP_MAX)]):
            layers[k] = layer
            keep.append(_keep)
        rc = fn(ctx, _ptr(page), h, w, 3, w * 3, layers, 3)
        return rc, [page]

    out['vkx_remap_u8'], out['vkx_remap_f32'], out['vkx_warp_affine_u8'] = remap_u8, remap_f32, warp
    out['vkx_grid_to_map'], out['vkx_grid_remap'], out['vkx_gaussian_blur_u8'] = grid_to_map, grid_remap, blur
    out['vkx_color_shift_rgb'], out['vkx_add_noise_i16'], out['vkx_line_streak_u8'], out['vkx_fill_u8'] = hue, add_noise, streak, fill
    return out, (img, score, sv, dv, (dh, dw), mx, my, noise, M)


def test_every_twin_takes_the_product_signature_and_gives_the_oracle_result():
    calls, (img, score, sv, dv, dshape, mx, my, noise, M) = _calls()
    assert sorted(calls) == sorted(TWINS)
    got = {name: calls[name](_twin(name), None) for name in TWINS}
    assert all(rc == 0 for rc, _ in got.values()), {k: v[0] for k, v in got.items()}
    assert (got['vkx_remap_u8'][1][0] == O.remap(img, mx, my)).all()
    assert (got['vkx_remap_f32'][1][0] == O.remap(score, mx, my)).all()
    assert (got['vkx_warp_affine_u8'][1][0] == O.warp_affine(img, M.reshape(2, 3), (img.shape[1] + 5, img.shape[0] + 9))).all()
    assert (got['vkx_grid_to_map'][1][0] == mx.view(np.uint32)).all() and (got['vkx_grid_to_map'][1][1] == my.view(np.uint32)).all()
    assert (got['vkx_grid_remap'][1][0] == O.remap(img, mx, my)).all()
    assert (got['vkx_gaussian_blur_u8'][1][0] == O.gaussian_blur(img, 5, 1.0)).all()
    assert (got['vkx_color_shift_rgb'][1][0] == O.color_shift_rgb(img, 37)).all()
    assert (got['vkx_add_noise_i16'][1][0] == O.add_noise_i16(img, noise)).all()
    assert (got['vkx_line_streak_u8'][1][0] == O.line_streak(img, 2, 9, 3, 5, (10, 200, 30), 0.6, True, True)).all()


def test_chain_twin_on_host_items():
    """``vkx_cpu_chain_rgb_batch``: the items of ``vkx_chain_rgb_batch_dev`` with host pointers."""
    rng = default_rng(9)
    fn = O.lib().vkx_cpu_chain_rgb_batch
    fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.POINTER(N.VkxChainItem), ctypes.c_int], ctypes.c_int
    items = (N.VkxChainItem * 2)()
    keep, want = [], []
    for k in range(2):
        h, w = 80 + 9 * k, 120 - 7 * k
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        sv, dv, (dh, dw) = _lattice(h, w, 20 + k)
        noise = np.round(default_rng(30 + k).normal(0, 8, (dh, dw, 3))).astype(np.int16)
        dst = np.zeros((dh, dw, 3), np.uint8)
        it = items[k]
        it.src, it.dst, it.src_stride, it.dst_stride = img.ctypes.data, dst.ctypes.data, w * 3, dw * 3
        it.sh, it.sw, it.dh, it.dw = h, w, dh, dw
        it.src_vertices, it.dst_vertices, it.rows, it.cols = sv.ctypes.data, dv.ctypes.data, sv.shape[0], sv.shape[1]
        it.noise, it.noise_stride_el = noise.ctypes.data, dw * 3
        if k == 0:
            it.blur_sigma, it.blur_ksize, it.hue_delta, it.hue_enabled = 1.0, 5, 37, 1
        mx, my = O.grid_to_map(sv, dv, (dh, dw))
        ref = O.remap(img, mx, my)
        if k == 0:
            ref = O.color_shift_rgb(O.gaussian_blur(ref, 5, 1.0), 37)
        want.append(O.add_noise_i16(ref, noise))
        keep.append((img, sv, dv, noise, dst))
    assert fn(None, items, 2) == 0
    for (_i, _s, _d, _n, dst), ref in zip(keep, want):
        assert (dst == ref).all()


@pytest.mark.gpu
def test_the_same_call_on_the_product_and_on_its_twin():
    ctx = N.default_ctx()
    calls, _ = _calls(seed=5)
    for name in TWINS:
        rc_gpu, out_gpu = calls[name](getattr(N.lib(), name), ctx.handle)
        rc_cpu, out_cpu = calls[name](_twin(name), None)
        assert rc_gpu == 0 and rc_cpu == 0, name
        for a, b in zip(out_gpu, out_cpu):
            assert (a == b).all(), name
