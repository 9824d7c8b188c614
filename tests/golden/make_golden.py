#!/usr/bin/env python3
"""Regenerates tests/golden/* by IMPORTING THE REFERENCE (read-only, /root/reference).

Run only in the build container (the reference never travels to the GPU box):

    python tests/golden/make_golden.py

The reference needs OpenCV & co., which are absent here, so the missing third-party modules
are stubbed with MagicMock.  Everything whose arithmetic is numpy-only
then runs for real and yields genuine reference outputs ("pinned" fixtures).  For two fixtures
the stubbed cv2 entry points are monkey-patched with this repo's oracle restatements so that
the reference's own *loop structure* (cell order, overwrite rule, vertex bookkeeping) is
exercised; those are marked ``oracle_patched`` and are structure checks, not independent pins.

Outputs are data only (inputs + expected outputs), never reference source text.
"""
import json
import os
import sys
from unittest.mock import MagicMock

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

for _m in ['cv2', 'iolite', 'shapely', 'shapely.geometry', 'shapely.strtree', 'shapely.validation', 'shapely.ops',
           'pyclipper', 'cattrs', 'cattrs.errors', 'cattrs.gen', 'intervaltree', 'faker', 'freetype', 'barcode',
           'rectpack', 'vkit_collect_usage_information', 'fireball', 'barcode.writer']:
    sys.modules[_m] = MagicMock(name=_m)
# the interpolation codes are data the reference passes around (utility/opt.py:125-148): give the stub cv2's integers
(sys.modules['cv2'].INTER_NEAREST, sys.modules['cv2'].INTER_LINEAR, sys.modules['cv2'].INTER_CUBIC,
 sys.modules['cv2'].INTER_AREA, sys.modules['cv2'].INTER_LANCZOS4, sys.modules['cv2'].INTER_LINEAR_EXACT,
 sys.modules['cv2'].INTER_NEAREST_EXACT) = range(7)
sys.path.insert(0, '/root/reference')

import attrs  # noqa: E402
import numpy as np  # noqa: E402
from numpy.random import default_rng  # noqa: E402

import cv2 as cv_stub  # noqa: E402  (the MagicMock)
from vkit.element import Image, Mask, ScoreMap, Box, Point, PointList, PointTuple, Polygon  # noqa: E402
from vkit.mechanism import distortion as D  # noqa: E402
from vkit.mechanism.distortion_policy import random_distortion as RD  # noqa: E402
from vkit.mechanism.distortion_policy.geometric import mls as P_mls, camera as P_cam, affine as P_aff  # noqa: E402
from vkit.mechanism.distortion_policy.photometric import (  # noqa: E402
    blur as P_blur, color as P_color, noise as P_noise, streak as P_streak, effect as P_effect,
)
from vkit.mechanism.distortion.geometric.mls import SimilarityMlsState  # noqa: E402
from vkit.mechanism.distortion.geometric import affine as G_aff  # noqa: E402
from vkit.mechanism.distortion.geometric import camera as G_cam  # noqa: E402
from vkit.mechanism.distortion.geometric.grid_rendering.grid_creator import create_src_image_grid  # noqa: E402

import oracle as O  # noqa: E402


def plain(obj):
    """Config object -> JSON-able structure (points as [smooth_y, smooth_x])."""
    if isinstance(obj, Point):
        return [obj.smooth_y, obj.smooth_x]
    if attrs.has(type(obj)):
        return {a.name.lstrip('_'): plain(getattr(obj, a.name)) for a in attrs.fields(type(obj))
                if a.name != '_rng_state'}
    if isinstance(obj, (list, tuple)):
        return [plain(v) for v in obj]
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.floating,)):
        return float(obj)
    if hasattr(obj, 'value') and obj.__class__.__module__.startswith('vkit'):
        return obj.value
    return obj


def grid_to_arrays(grid):
    smooth = np.asarray([[(p.smooth_x, p.smooth_y) for p in row] for row in grid.points_2d], dtype=np.float64)
    ints = np.asarray([[(p.x, p.y) for p in row] for row in grid.points_2d], dtype=np.int32)
    return smooth, ints


# --------------------------------------------------------------------------------------------
def gen_numpy_path():
    out = {}
    rng = default_rng(42)

    # B.2 known answer.
    bg = Image(mat=np.full((4, 4, 3), 200, np.uint8))
    sm = ScoreMap(mat=np.linspace(0, 1, 16, dtype=np.float32).reshape(4, 4))
    sm.fill_image(bg, (10, 20, 30))
    out['fill_ka_out'] = bg.mat.copy()

    # ScoreMap-alpha constant-colour layers (text lines), box attached.
    page = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    out['fill_sm_in'] = page.copy()
    alpha = rng.random((12, 30), dtype=np.float32)
    alpha[alpha < 0.4] = 0
    alpha[0, 0] = 1.0
    box = Box(up=5, down=16, left=20, right=49)
    img = Image(mat=page.copy())
    ScoreMap(mat=alpha, box=box).fill_image(img, (10, 20, 30))
    out['fill_sm_alpha'] = alpha
    out['fill_sm_box'] = np.asarray([5, 20, 12, 30])
    out['fill_sm_out'] = img.mat.copy()

    # Mask fill with an image value (inactive-region fill) and with scalar alpha.
    m = (rng.random((40, 56)) < 0.5).astype(np.uint8)
    val = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    img = Image(mat=page.copy())
    Mask(mat=m).fill_image(img, Image(mat=val))
    out['fill_mask_mask'] = m
    out['fill_mask_value'] = val
    out['fill_mask_out'] = img.mat.copy()
    for a in (0.3, 0.5, 0.999, 1.0, 0.0):
        img = Image(mat=page.copy())
        Mask(mat=m).fill_image(img, (7, 99, 250), alpha=a)
        out[f'fill_mask_const_a{a}'] = img.mat.copy()
        img = Image(mat=page.copy())
        Mask(mat=m).fill_image(img, Image(mat=val), alpha=a)
        out[f'fill_mask_img_a{a}'] = img.mat.copy()

    # Box fill: image value with scalar alpha (page images / symbols), and alpha array + mask.
    sub = rng.integers(0, 256, (10, 17, 3), dtype=np.uint8)
    out['fill_box_value'] = sub
    for a in (0.25, 1.0):
        img = Image(mat=page.copy())
        Box(up=3, down=12, left=8, right=24).fill_image(img, Image(mat=sub), alpha=a)
        out[f'fill_box_img_a{a}'] = img.mat.copy()
    bmask = (rng.random((10, 17)) < 0.6).astype(np.uint8)
    balpha = rng.random((10, 17), dtype=np.float32)
    img = Image(mat=page.copy())
    Box(up=3, down=12, left=8, right=24).fill_image(img, (200, 10, 10), image_mask=Mask(mat=bmask), alpha=0.7)
    out['fill_box_mask'] = bmask
    out['fill_box_mask_out'] = img.mat.copy()
    img = Image(mat=page.copy())
    Box(up=3, down=12, left=8, right=24).fill_image(img, (200, 10, 10), alpha=ScoreMap(mat=balpha))
    out['fill_box_alpha'] = balpha
    out['fill_box_alpha_out'] = img.mat.copy()

    # Photometric numpy-only members.
    src = default_rng(0).integers(0, 256, (24, 31, 3), dtype=np.uint8)
    out['photo_src'] = src
    im = Image(mat=src.copy())
    out['noise_std10_seed1'] = D.gaussion_noise.distort_image(D.GaussionNoiseConfig(std=10), im, rng=default_rng(1)).mat
    out['noise_std33_seed2'] = D.gaussion_noise.distort_image(D.GaussionNoiseConfig(std=33.3), im, rng=default_rng(2)).mat
    out['noise_std10_seed1_plane'] = np.round(default_rng(1).normal(0, 10, src.shape)).astype(np.int16)
    out['mean_shift_100'] = D.mean_shift.distort_image(D.MeanShiftConfig(delta=100), im).mat
    out['mean_shift_m60_c1'] = D.mean_shift.distort_image(D.MeanShiftConfig(delta=-60, channels=[1]), im).mat
    out['mean_shift_128_cycle_thr'] = D.mean_shift.distort_image(
        D.MeanShiftConfig(delta=128, oob_behavior=D.OutOfBoundBehavior.CYCLE, threshold=127), im).mat
    out['mean_shift_m128_cycle_thr'] = D.mean_shift.distort_image(
        D.MeanShiftConfig(delta=-128, oob_behavior=D.OutOfBoundBehavior.CYCLE, threshold=128), im).mat
    out['mean_shift_40_thr200'] = D.mean_shift.distort_image(D.MeanShiftConfig(delta=40, threshold=200), im).mat
    out['line_streak_a03'] = D.line_streak.distort_image(D.LineStreakConfig(alpha=0.3), im).mat
    out['line_streak_dash'] = D.line_streak.distort_image(
        D.LineStreakConfig(thickness=2, gap=5, dash_thickness=3, dash_gap=2, color=(9, 8, 7), alpha=0.5), im).mat
    out['line_streak_vert_a1'] = D.line_streak.distort_image(
        D.LineStreakConfig(thickness=1, gap=3, alpha=1.0, enable_hori=False, color=(1, 2, 3)), im).mat
    out['rect_streak'] = D.rectangle_streak.distort_image(
        D.RectangleStreakConfig(thickness=2, short_side_min=4, short_side_step=5, alpha=0.6, color=(5, 6, 7)), im).mat
    out['rect_streak_dash'] = D.rectangle_streak.distort_image(
        D.RectangleStreakConfig(thickness=1, aspect_ratio=0.7, short_side_min=3, short_side_step=4, dash_thickness=2,
                                dash_gap=1, alpha=1.0), im).mat
    # hue add on an HSV-mode image (no cvtColor involved)
    from vkit.element import ImageMode
    hsv = Image(mat=src.copy(), mode=ImageMode.HSV)
    out['color_shift_hsv_37'] = D.color_shift.distort_image(D.ColorShiftConfig(delta=37), hsv).mat
    out['color_shift_hsv_m200'] = D.color_shift.distort_image(D.ColorShiftConfig(delta=-200), hsv).mat
    np.savez_compressed(os.path.join(HERE, 'numpy_path.npz'), **out)


# --------------------------------------------------------------------------------------------
def gen_fill_modes():
    """fill_np_array on float32 score maps and with keep_max_value / keep_min_value (label rasterisation)."""
    out = {}
    rng = default_rng(77)
    sm0 = (rng.random((24, 40), dtype=np.float32) * 30).astype(np.float32)
    m = (rng.random((24, 40)) < 0.45).astype(np.uint8)
    val = (rng.random((24, 40), dtype=np.float32) * 30).astype(np.float32)
    out['f32_in'], out['f32_mask'], out['f32_value'] = sm0, m, val
    for tag, kwargs in [('plain', {}), ('max', dict(keep_max_value=True)), ('min', dict(keep_min_value=True))]:
        sm = ScoreMap(mat=sm0.copy(), is_prob=False)
        Mask(mat=m).fill_score_map(sm, 12.5, **kwargs)
        out[f'f32_mask_const_{tag}'] = sm.mat.copy()
        sm = ScoreMap(mat=sm0.copy(), is_prob=False)
        Mask(mat=m).fill_score_map(sm, ScoreMap(mat=val, is_prob=False), **kwargs)
        out[f'f32_mask_plane_{tag}'] = sm.mat.copy()
        sm = ScoreMap(mat=sm0.copy(), is_prob=False)
        Box(up=2, down=19, left=5, right=33).fill_score_map(sm, 7.25, **kwargs)
        out[f'f32_box_const_{tag}'] = sm.mat.copy()
    out['f32_box'] = np.asarray([2, 5, 18, 29])
    # uint8 destinations with the keep modes (Mask.fill_mask / fill_np_array on an image plane)
    mk0 = rng.integers(0, 4, (24, 40)).astype(np.uint8)
    mv = rng.integers(0, 4, (24, 40)).astype(np.uint8)
    out['u8_in'], out['u8_value'] = mk0, mv
    for tag, kwargs in [('max', dict(keep_max_value=True)), ('min', dict(keep_min_value=True))]:
        mk = Mask(mat=mk0.copy())
        Mask(mat=m).fill_mask(mk, 2, **kwargs)
        out[f'u8_mask_const_{tag}'] = mk.mat.copy()
        mk = Mask(mat=mk0.copy())
        Mask(mat=m).fill_mask(mk, mv, **kwargs)
        out[f'u8_mask_plane_{tag}'] = mk.mat.copy()
    # float32 blends (alpha array / scalar) through the raw function
    from vkit.element.opt import fill_np_array
    alpha = (rng.random((24, 40), dtype=np.float32) * (rng.random((24, 40)) < 0.6)).astype(np.float32)
    out['f32_alpha'] = alpha
    dst = sm0.copy()
    fill_np_array(dst, val, np_mask=alpha > 0, alpha=alpha)
    out['f32_alpha_plane'] = dst
    dst = sm0.copy()
    fill_np_array(dst, 3.0, alpha=0.3)
    out['f32_alpha_scalar'] = dst
    np.savez_compressed(os.path.join(HERE, 'fill_modes.npz'), **out)


def gen_pointwise_ops():
    """complement / posterization / channel_permutation / impulse_noise / speckle_noise: numpy-only members."""
    out = {}
    src = default_rng(91).integers(0, 256, (37, 53, 3), dtype=np.uint8)
    out['src'] = src
    img = Image(mat=src)
    cases = [dict(), dict(threshold=100), dict(threshold=100, enable_threshold_lte=True),
             dict(threshold=0, channels=[1]), dict(threshold=255, enable_threshold_lte=True, channels=[0, 2]),
             dict(channels=[2])]
    for i, kw in enumerate(cases):
        out[f'complement_{i}'] = D.complement.distort(D.ComplementConfig(**kw), image=img).image.mat
    out['complement_cases'] = np.asarray(json.dumps(cases))
    for bits in range(8):
        out[f'posterization_{bits}'] = D.posterization.distort(D.PosterizationConfig(num_bits=bits), image=img).image.mat
    out['posterization_3_c1'] = D.posterization.distort(D.PosterizationConfig(num_bits=3, channels=[1]), image=img).image.mat
    for seed in (0, 1, 2, 3):
        out[f'channel_permutation_{seed}'] = D.channel_permutation.distort(D.ChannelPermutationConfig(), image=img, rng=default_rng(seed)).image.mat
    for i, (ps, pp, seed) in enumerate([(0.02, 0.03, 0), (0.0, 0.05, 1), (0.3, 0.0, 2), (0.0, 0.0, 3)]):
        out[f'impulse_{i}'] = D.impulse_noise.distort(D.ImpulseNoiseConfig(prob_salt=ps, prob_pepper=pp), image=img,
                                                      rng=default_rng(seed)).image.mat
    out['impulse_cases'] = np.asarray([[0.02, 0.03, 0], [0.0, 0.05, 1], [0.3, 0.0, 2], [0.0, 0.0, 3]])
    for i, (std, seed) in enumerate([(0.1, 0), (0.3, 1), (1.5, 2)]):
        out[f'speckle_{i}'] = D.speckle_noise.distort(D.SpeckleNoiseConfig(std=std), image=img, rng=default_rng(seed)).image.mat
    out['speckle_cases'] = np.asarray([[0.1, 0], [0.3, 1], [1.5, 2]])
    low = (src // 3 + 40).astype(np.uint8)          # a narrow value range, so the equalisation has work to do
    low[:, :, 1] = 77                                 # one flat channel (delta == 0: left alone)
    out['beq_src'] = low
    limg = Image(mat=low)
    out['beq_all'] = D.boundary_equalization.distort(D.BoundaryEqualizationConfig(), image=limg).image.mat
    out['beq_c02'] = D.boundary_equalization.distort(D.BoundaryEqualizationConfig(channels=[0, 2]), image=limg).image.mat
    out['beq_c1'] = D.boundary_equalization.distort(D.BoundaryEqualizationConfig(channels=[1]), image=limg).image.mat
    out['beq_gray'] = D.boundary_equalization.distort(D.BoundaryEqualizationConfig(),
                                                       image=Image(mat=low[:, :, 0].copy())).image.mat
    for i, (rough, rmax, rmin, seed) in enumerate([(0.5, 1.0, 0.0, 0), (0.85, 0.6, 0.1, 1), (0.2, 0.3, 0.0, 2)]):
        out[f'fog_{i}'] = D.fog.distort(D.FogConfig(roughness=rough, ratio_max=rmax, ratio_min=rmin), image=img,
                                        rng=default_rng(seed)).image.mat
    out['fog_cases'] = np.asarray([[0.5, 1.0, 0.0, 0], [0.85, 0.6, 0.1, 1], [0.2, 0.3, 0.0, 2]])
    wide = Image(mat=default_rng(92).integers(0, 256, (70, 300, 3), dtype=np.uint8))
    out['fog_wide_src'] = wide.mat
    out['fog_wide'] = D.fog.distort(D.FogConfig(roughness=0.6, fog_rgb=(200, 10, 30)), image=wide, rng=default_rng(7)).image.mat
    # glass_blur's pixel shuffle: the blur is stubbed with the identity and the image encodes its own coordinates, so
    # the output IS the (row, column) source plane pair of the reference's numpy bookkeeping
    coords = np.zeros((97, 141, 3), np.uint8)
    coords[:, :, 0] = np.arange(97).reshape(-1, 1)
    coords[:, :, 1] = np.arange(141).reshape(1, -1)
    saved_blur = cv_stub.GaussianBlur
    cv_stub.GaussianBlur = lambda mat, ksize, sigma: mat
    for i, (delta, loop, seed) in enumerate([(1, 1, 0), (1, 4, 1), (2, 5, 2), (3, 2, 3)]):
        out[f'glass_planes_{i}'] = D.glass_blur.distort(D.GlassBlurConfig(sigma=1.0, delta=delta, loop=loop),
                                                         image=Image(mat=coords), rng=default_rng(seed)).image.mat[:, :, :2]
    out['glass_cases'] = np.asarray([(1, 1, 0), (1, 4, 1), (2, 5, 2), (3, 2, 3)])
    cv_stub.GaussianBlur = saved_blur
    for seed in (0, 1):
        out[f'poisson_{seed}'] = D.poisson_noise.distort(D.PoissonNoiseConfig(), image=img, rng=default_rng(seed)).image.mat
    # zoom_in_blur: cv.resize is substituted by the oracle's bicubic restatement (oracle_patched: a structure check of
    # the numpy part -- step list, uint16 accumulation, float64 blend and rounding)
    saved_resize = cv_stub.resize
    cv_stub.resize = lambda mat, dsize, interpolation=None: O.resize_cubic(mat, (dsize[1], dsize[0]))
    for i, (ratio, step, alpha) in enumerate([(0.1, 0.01, 0.5), (0.05, 0.02, 0.7), (0.033, 0.004, 0.61)]):
        out[f'zoom_oracle_patched_{i}'] = D.zoom_in_blur.distort(D.ZoomInBlurConfig(ratio=ratio, step=step, alpha=alpha),
                                                                 image=img).image.mat
    out['zoom_cases'] = np.asarray([(0.1, 0.01, 0.5), (0.05, 0.02, 0.7), (0.033, 0.004, 0.61)])
    cv_stub.resize = saved_resize
    gray = Image(mat=src[:, :, 0].copy())
    out['gray_complement_thr'] = D.complement.distort(D.ComplementConfig(threshold=128), image=gray).image.mat
    out['gray_impulse'] = D.impulse_noise.distort(D.ImpulseNoiseConfig(prob_salt=0.1, prob_pepper=0.1), image=gray,
                                                  rng=default_rng(5)).image.mat
    out['gray_speckle'] = D.speckle_noise.distort(D.SpeckleNoiseConfig(std=0.2), image=gray, rng=default_rng(6)).image.mat
    np.savez_compressed(os.path.join(HERE, 'pointwise_ops.npz'), **out)


def gen_mls_states():
    out = {}
    cases = [(64, 64, 0, 5), (96, 80, 1, 8), (130, 257, 2, 10), (512, 512, 0, 5), (300, 200, 3, 1)]
    meta = []
    for (h, w, seed, level) in cases:
        cfg = P_mls.SimilarityMlsConfigGenerator(P_mls.SimilarityMlsConfigGeneratorConfig(), level)((h, w), default_rng(seed))
        st = SimilarityMlsState(cfg, (h, w), None)
        key = f'{h}x{w}_s{seed}_l{level}'
        out[key + '_src_handles'] = np.asarray([[p.smooth_x, p.smooth_y] for p in cfg.src_handle_points])
        out[key + '_dst_handles'] = np.asarray([[p.smooth_x, p.smooth_y] for p in cfg.dst_handle_points])
        s_smooth, s_int = grid_to_arrays(st.src_image_grid)
        d_smooth, d_int = grid_to_arrays(st.dst_image_grid)
        out[key + '_src_grid'] = s_int
        out[key + '_dst_grid_smooth'] = d_smooth
        out[key + '_dst_grid'] = d_int
        meta.append(dict(key=key, h=h, w=w, seed=seed, level=level, grid_size=cfg.grid_size,
                         result_shape=list(st.result_shape),
                         shift=[st.shift_amount_y, st.shift_amount_x]))
    # Known answers for the BASELINE sizes (vertex [1][1] and result shape only; full grids are large).
    for hw in (2048, 4096):
        cfg = P_mls.SimilarityMlsConfigGenerator(P_mls.SimilarityMlsConfigGeneratorConfig(), 5)((hw, hw), default_rng(0))
        st = SimilarityMlsState(cfg, (hw, hw), None)
        p = st.dst_image_grid.points_2d[1][1]
        d_smooth, d_int = grid_to_arrays(st.dst_image_grid)
        meta.append(dict(key=f'big{hw}', h=hw, w=hw, seed=0, level=5, grid_size=cfg.grid_size,
                         result_shape=list(st.result_shape), dst11=[p.smooth_y, p.smooth_x],
                         grid_checksum=int(np.asarray(d_int, dtype=np.int64).sum()),
                         rows=len(st.dst_image_grid.points_2d), cols=len(st.dst_image_grid.points_2d[0])))
    out['meta_json'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'mls_states.npz'), **out)


def gen_mls_lattices():
    """Full destination lattices of the real reference at the BASELINE sizes (C2 / C5: 2048^2, 4096^2) and the page
    size of C4 (1024^2): handle points in, rounded + smooth lattice out.  The per-vertex reference loop takes ~0.8 s a
    state here; the fixtures pin the device kernel (csrc/mls.hip) and the oracle restatement at those sizes."""
    out = {}
    meta = []
    for (hw, seed, level) in ((1024, 0, 5), (1024, 3, 9), (2048, 0, 5), (2048, 1, 5), (2048, 2, 10), (4096, 0, 5),
                              (4096, 1, 5)):
        cfg = P_mls.SimilarityMlsConfigGenerator(P_mls.SimilarityMlsConfigGeneratorConfig(), level)((hw, hw), default_rng(seed))
        st = SimilarityMlsState(cfg, (hw, hw), None)
        key = f'{hw}_s{seed}_l{level}'
        out[key + '_src_handles'] = np.asarray([[p.smooth_x, p.smooth_y] for p in cfg.src_handle_points])
        out[key + '_dst_handles'] = np.asarray([[p.smooth_x, p.smooth_y] for p in cfg.dst_handle_points])
        d_smooth, d_int = grid_to_arrays(st.dst_image_grid)
        out[key + '_dst_grid'] = d_int.astype(np.int16)
        # the smooth lattice is (float32 projection) - (integer shift): stored as the float32 projection, exactly
        unshifted = d_smooth + np.asarray([st.shift_amount_x, st.shift_amount_y], dtype=np.float64)
        out[key + '_projected'] = unshifted.astype(np.float32)
        assert (out[key + '_projected'].astype(np.float64) - np.asarray([st.shift_amount_x, st.shift_amount_y]) == d_smooth).all()
        meta.append(dict(key=key, h=hw, w=hw, seed=seed, level=level, grid_size=cfg.grid_size,
                         result_shape=list(st.result_shape), shift=[st.shift_amount_y, st.shift_amount_x]))
    out['meta_json'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'mls_lattices.npz'), **out)


# --------------------------------------------------------------------------------------------
def gen_affine_states():
    out = []
    for (h, w) in [(512, 512), (567, 440), (31, 77)]:
        for angle in [0, 1, 30, 45, 90, 91, 132, 180, 200, 269, 270, 271, 359, 360, -30, 725]:
            st = G_aff.RotateState(G_aff.RotateConfig(angle), (h, w), None)
            out.append(dict(kind='rotate', h=h, w=w, angle=angle, trans_mat=st.trans_mat.astype(np.float64).tolist(),
                            dsize=list(st.dsize)))
        for angle in [-30, -1, 0, 1, 17, 30]:
            for kind, cls, cfgcls in (('shear_hori', G_aff.ShearHoriState, G_aff.ShearHoriConfig),
                                      ('shear_vert', G_aff.ShearVertState, G_aff.ShearVertConfig)):
                st = cls(cfgcls(angle), (h, w), None)
                out.append(dict(kind=kind, h=h, w=w, angle=angle,
                                trans_mat=None if st.trans_mat is None else st.trans_mat.astype(np.float64).tolist(),
                                dsize=None if st.dsize is None else list(st.dsize)))
    # rotate applied to points (numpy only)
    st = G_aff.RotateState(G_aff.RotateConfig(30), (512, 512), None)
    pts = PointTuple(Point.create(y=y, x=x) for y, x in [(0, 0), (10.5, 20.25), (511, 511), (100, 3)])
    new = G_aff.affine_points(st.trans_mat, pts)
    out.append(dict(kind='rotate_points', angle=30, h=512, w=512, src=[[p.smooth_y, p.smooth_x] for p in pts],
                    dst=[[p.smooth_y, p.smooth_x] for p in new]))
    with open(os.path.join(HERE, 'affine_states.json'), 'w') as f:
        json.dump(out, f)


# --------------------------------------------------------------------------------------------
def gen_policy_configs():
    gens = {
        'similarity_mls': (P_mls.SimilarityMlsConfigGenerator, P_mls.SimilarityMlsConfigGeneratorConfig),
        'camera_plane_only': (P_cam.CameraPlaneOnlyConfigGenerator, P_cam.CameraPlaneOnlyConfigGeneratorConfig),
        'camera_cubic_curve': (P_cam.CameraCubicCurveConfigGenerator, P_cam.CameraCubicCurveConfigGeneratorConfig),
        'camera_plane_line_fold': (P_cam.CameraPlaneLineFoldConfigGenerator, P_cam.CameraPlaneLineFoldConfigGeneratorConfig),
        'camera_plane_line_curve': (P_cam.CameraPlaneLineCurveConfigGenerator, P_cam.CameraPlaneLineCurveConfigGeneratorConfig),
        'shear_hori': (P_aff.ShearHoriConfigGenerator, P_aff.ShearHoriConfigGeneratorConfig),
        'shear_vert': (P_aff.ShearVertConfigGenerator, P_aff.ShearVertConfigGeneratorConfig),
        'rotate': (P_aff.RotateConfigGenerator, P_aff.RotateConfigGeneratorConfig),
        'skew_hori': (P_aff.SkewHoriConfigGenerator, P_aff.SkewHoriConfigGeneratorConfig),
        'skew_vert': (P_aff.SkewVertConfigGenerator, P_aff.SkewVertConfigGeneratorConfig),
        'gaussian_blur': (P_blur.GaussianBlurConfigGenerator, P_blur.GaussianBlurConfigGeneratorConfig),
        'defocus_blur': (P_blur.DefocusBlurConfigGenerator, P_blur.DefocusBlurConfigGeneratorConfig),
        'motion_blur': (P_blur.MotionBlurConfigGenerator, P_blur.MotionBlurConfigGeneratorConfig),
        'mean_shift': (P_color.MeanShiftConfigGenerator, P_color.MeanShiftConfigGeneratorConfig),
        'color_shift': (P_color.ColorShiftConfigGenerator, P_color.ColorShiftConfigGeneratorConfig),
        'gaussion_noise': (P_noise.GaussionNoiseConfigGenerator, P_noise.GaussionNoiseConfigGeneratorConfig),
        'impulse_noise': (P_noise.ImpulseNoiseConfigGenerator, P_noise.ImpulseNoiseConfigGeneratorConfig),
        'speckle_noise': (P_noise.SpeckleNoiseConfigGenerator, P_noise.SpeckleNoiseConfigGeneratorConfig),
        'complement': (P_color.ComplementConfigGenerator, P_color.ComplementConfigGeneratorConfig),
        'glass_blur': (P_blur.GlassBlurConfigGenerator, P_blur.GlassBlurConfigGeneratorConfig),
        'zoom_in_blur': (P_blur.ZoomInBlurConfigGenerator, P_blur.ZoomInBlurConfigGeneratorConfig),
        'fog': (P_effect.FogConfigGenerator, P_effect.FogConfigGeneratorConfig),
        'pixelation': (P_effect.PixelationConfigGenerator, P_effect.PixelationConfigGeneratorConfig),
        'boundary_equalization': (P_color.BoundaryEqualizationConfigGenerator,
                                  P_color.BoundaryEqualizationConfigGeneratorConfig),
        'histogram_equalization': (P_color.HistogramEqualizationConfigGenerator,
                                   P_color.HistogramEqualizationConfigGeneratorConfig),
        'brightness_shift': (P_color.BrightnessShiftConfigGenerator, P_color.BrightnessShiftConfigGeneratorConfig),
        'std_shift': (P_color.StdShiftConfigGenerator, P_color.StdShiftConfigGeneratorConfig),
        'color_balance': (P_color.ColorBalanceConfigGenerator, P_color.ColorBalanceConfigGeneratorConfig),
        'posterization': (P_color.PosterizationConfigGenerator, P_color.PosterizationConfigGeneratorConfig),
        'channel_permutation': (P_color.ChannelPermutationConfigGenerator,
                                P_color.ChannelPermutationConfigGeneratorConfig),
        'line_streak': (P_streak.LineStreakConfigGenerator, P_streak.LineStreakConfigGeneratorConfig),
        'rectangle_streak': (P_streak.RectangleStreakConfigGenerator, P_streak.RectangleStreakConfigGeneratorConfig),
        'ellipse_streak': (P_streak.EllipseStreakConfigGenerator, P_streak.EllipseStreakConfigGeneratorConfig),
        'jpeg_quality': (P_effect.JpegQualityConfigGenerator, P_effect.JpegQualityConfigGeneratorConfig),
    }
    out = []
    for name, (gen_cls, cfg_cls) in gens.items():
        for level in (1, 5, 10) if name != 'similarity_mls' else (1, 5):
            for seed in (0, 1, 2):
                for shape in ((96, 80), (2048, 2048)) if name != 'similarity_mls' else ((96, 80),):
                    rng = default_rng(seed)
                    cfg = gen_cls(cfg_cls(), level)(shape, rng)
                    out.append(dict(name=name, level=level, seed=seed, shape=list(shape), config=plain(cfg),
                                    next_random=float(rng.random())))
    # Known answer quoted in SURVEY 8(d): seed 0 @2048^2 cubic curve.
    with open(os.path.join(HERE, 'policy_configs.json'), 'w') as f:
        json.dump(out, f)


# --------------------------------------------------------------------------------------------
def gen_operator_semantics():
    out = {}
    img = Image(mat=default_rng(0).integers(0, 256, (16, 16, 3), dtype=np.uint8))
    m = Mask(mat=(default_rng(3).random((16, 16)) < 0.5).astype(np.uint8))
    s = ScoreMap(mat=default_rng(4).random((16, 16), dtype=np.float32))
    rng = default_rng(7)
    res = P_noise.gaussion_noise_policy_factory.create(None).distort(
        level=5, image=img, mask=m, score_map=s, rng=rng, enable_debug=True)
    out['noise_policy'] = dict(
        std=res.config.std, mask_is_same=res.mask is m, score_map_is_same=res.score_map is s,
        state_is_none=res.state is None, shape=list(res.shape), px00=res.image.mat[0, 0].tolist(),
        image_sum=int(res.image.mat.astype(np.int64).sum()), next_random=float(rng.random()),
        mode=res.image.mode.value,
    )
    again = D.gaussion_noise.distort_image(res.config, img)
    out['noise_policy']['replay_equal'] = bool((again.mat == res.image.mat).all())

    # geometric op through the operator: rotate with points / polygon / clipping (numpy only parts).
    rng = default_rng(11)
    pts = PointList([Point.create(y=1.5, x=2.5), Point.create(y=15, x=15), Point.create(y=0, x=15)])
    poly = Polygon.create(points=[Point.create(y=2, x=2), Point.create(y=2, x=12), Point.create(y=12, x=12)])
    res = D.rotate.distort(D.RotateConfig(angle=33), shapable_or_shape=(16, 16), points=pts,
                           polygon=poly, corner_points=pts, get_state=True)
    out['rotate_op'] = dict(
        shape=list(res.shape), points=[[p.smooth_y, p.smooth_x] for p in res.points],
        corner_points=[[p.smooth_y, p.smooth_x] for p in res.corner_points],
        polygon=[[p.smooth_y, p.smooth_x] for p in res.polygon.points])
    with open(os.path.join(HERE, 'operator_semantics.json'), 'w') as f:
        json.dump(out, f)


# --------------------------------------------------------------------------------------------
def gen_random_distortion_sampling():
    rd = RD.random_distortion_factory.create(None)
    out = dict(stage_sizes=[len(s.config.distortion_policies) for s in rd.stages],
               stage_names=[[p.name for p in s.config.distortion_policies] for s in rd.stages],
               stage_probs=[list(map(float, s.distortion_policy_probs)) for s in rd.stages],
               prob_enable=[s.config.prob_enable for s in rd.stages],
               samples=[])
    for seed in range(12):
        rng = default_rng(seed)
        rec = []
        for s in rd.stages:
            pols = s.sample_distortion_policies(rng)
            rec.append([p.name for p in pols])
        out['samples'].append(dict(seed=seed, names=rec, next_random=float(rng.random())))
    rd2 = RD.random_distortion_factory.create(RD.RandomDistortionFactoryConfig(
        force_post_rotate=True, disabled_policy_names=['defocus_blur', 'zoom_in_blur']))
    out['post_rotate_stage_names'] = [[p.name for p in s.config.distortion_policies] for s in rd2.stages]
    with open(os.path.join(HERE, 'random_distortion_sampling.json'), 'w') as f:
        json.dump(out, f)


# --------------------------------------------------------------------------------------------
def gen_std_shift():
    """std_shift of the reference (photometric/color.py:165-210; numpy only): colour and grayscale images, channel
    subsets, scales on both sides of 1, and a plane big enough for the float32 running sum to leave the exact range."""
    out, cases = {}, []
    rng = default_rng(321)
    specs = [((37, 53, 3), 1.7, None), ((37, 53, 3), 0.45, [0, 2]), ((37, 53, 3), 2.5, [1]), ((64, 48), 1.3, None),
             ((64, 48), 0.8, None), ((1, 1, 3), 2.0, None),
             # formula-generated planes, large enough for the float32 running sums to leave the exact range
             ((5000, 4000), 1.21, None), ((1200, 1000, 3), 1.9, None), ((1200, 1000, 3), 0.6, [0, 2])]
    for i, (shape, scale, channels) in enumerate(specs):
        if shape[0] >= 1000:
            n = int(np.prod(shape))
            mat = (np.arange(n, dtype=np.uint32) * np.uint32(2654435761) >> np.uint32(24)).astype(np.uint8).reshape(shape)
            keep = None        # regenerated by the test from the same formula; only a digest of the output is kept
        else:
            mat = rng.integers(0, 256, shape, dtype=np.uint8)
            keep = mat
        res = D.std_shift.distort(D.StdShiftConfig(scale=scale, channels=channels), image=Image(mat=mat)).image.mat
        cases.append({'shape': list(shape), 'scale': scale, 'channels': channels})
        if keep is not None:
            out[f'in_{i}'] = keep
            out[f'out_{i}'] = res
        else:
            hist = np.bincount(res.reshape(-1), minlength=256)
            out[f'out_hist_{i}'] = hist
            out[f'out_head_{i}'] = res.reshape(-1)[:4096].copy()
    out['cases_json'] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'std_shift.npz'), **out)


def gen_gcn():
    """The float32 ``*_GCN`` image modes (vkit/element/image.py:733-768; numpy only).  ``to_gcn_image`` cannot succeed in the
    reference -- ``ImageMode.supports_gcn_mode`` is inverted (image.py:69-72), so every mode that has a GCN twin raises and every
    other one fails the table look-up -- which is recorded here as behaviour; a GCN image therefore only exists when a caller builds
    one (``Image(mat=float32, mode=RGB_GCN)``), and what the reference does with it on this path is ``to_non_gcn_image``: the
    min / max rescale to uint8 whose outputs are kept."""
    from vkit.element.image import ImageMode
    out, cases = {}, []
    rng = default_rng(77)
    raised = {}
    for mode in (ImageMode.RGB, ImageMode.GRAYSCALE, ImageMode.HSV, ImageMode.HSL, ImageMode.RGBA):
        shape = (5, 4) if mode == ImageMode.GRAYSCALE else (5, 4, 4 if mode == ImageMode.RGBA else 3)
        try:
            Image(mat=np.zeros(shape, np.uint8), mode=mode).to_gcn_image()
            raised[mode.value] = None
        except Exception as exc:
            raised[mode.value] = type(exc).__name__
    specs = [((41, 37, 3), 'rgb_gcn', 1.0, 0.0), ((33, 64), 'grayscale_gcn', 3.5, -2.0), ((57, 43, 3), 'hsv_gcn', 0.01, 7.0),
             ((57, 43, 3), 'hsl_gcn', 1000.0, 0.0), ((2, 2, 3), 'rgb_gcn', 1.0, 0.0)]
    for i, (shape, mode, spread, offset) in enumerate(specs):
        mat = (rng.standard_normal(shape) * spread + offset).astype(np.float32)
        back = Image(mat=mat, mode=ImageMode(mode)).to_non_gcn_image()
        cases.append({'shape': list(shape), 'mode': mode, 'back_mode': back.mode.value})
        out[f'in_{i}'] = mat
        out[f'back_{i}'] = back.mat
    out['cases_json'] = np.frombuffer(json.dumps({'cases': cases, 'to_gcn_image_raises': raised}).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'gcn.npz'), **out)


def gen_page_resizing():
    """PageResizingStep.run of the reference on recording stand-ins for the page elements: which size and which
    interpolation every element is asked to take, for seeded rngs (pipeline/text_detection/page_resizing.py:86-181)."""
    from vkit.pipeline.text_detection import page_resizing as PR
    from vkit.utility.opt import sample_cv_resize_interpolation
    fields = ['page_image', 'page_active_mask', 'page_char_mask', 'page_seal_impression_char_mask',
              'page_char_height_score_map', 'page_text_line_mask', 'page_text_line_height_score_map']

    class Recorder:
        def __init__(self, name, shape, log):
            self.name, self.shape, self.log = name, shape, log
            self.mat = 1.0

        def _resized(self, **kwargs):
            self.log.append([self.name, kwargs['resized_height'], kwargs['resized_width'], kwargs['cv_resize_interpolation']])
            return Recorder(self.name, (kwargs['resized_height'], kwargs['resized_width']), self.log)

        to_resized_image = to_resized_mask = to_resized_score_map = _resized

        def assign_mat(self, mat):
            self.log.append([self.name + '.scale', float(mat)])

    cases = []
    rng = default_rng(99)
    for seed in range(40):
        shape = (int(rng.integers(200, 1500)), int(rng.integers(200, 1500)))
        heights = [float(v) for v in rng.uniform(0.2, 60.0, int(rng.integers(3, 40)))]
        if seed % 5 == 0:
            heights.append(400.0)          # an outlier the median filter drops
        cfg = PR.PageResizingStepConfig(resized_text_line_height_min=3.0 + seed % 4, resized_text_line_height_max=10.0 + seed)
        step = PR.PageResizingStep(cfg)
        log = []
        page = MagicMock()
        for name in fields:
            setattr(page, name, Recorder(name, shape, log))
        page.page_text_line_heights = heights
        step.run(PR.PageResizingStepInput(page_distortion_step_output=page), default_rng(seed))
        cases.append({'seed': seed, 'shape': list(shape), 'heights': heights,
                      'config': [cfg.resized_text_line_height_min, cfg.resized_text_line_height_max,
                                 cfg.text_line_heights_filtering_thr],
                      'heights_min': step.get_text_line_heights_min(heights), 'calls': log})
    draws = [[int(sample_cv_resize_interpolation(default_rng(s), bool(a))) for a in (0, 1)] for s in range(64)]
    with open(os.path.join(HERE, 'page_resizing.json'), 'w') as f:
        json.dump({'cases': cases, 'interpolation_draws': draws}, f)


def _patch_cv2_with_oracle():
    def gpt(a, b, flag=None):
        return O.get_perspective_transform(a, b, O.SOLVER_HYBRID)

    def fill_poly(img, pts_list, color):
        assert color == 1 and len(pts_list) == 1
        m = O.fill_poly(img.shape, pts_list[0])
        img[m > 0] = 1
        return img

    def rodrigues(rvec):
        R = O.rodrigues(np.asarray(rvec, dtype=np.float64))
        return R.astype(np.asarray(rvec).dtype), None

    def project_points(p3, rvec, tvec, K, dist):
        out = O.project_points(np.asarray(p3, np.float64), np.asarray(rvec, np.float64).reshape(3),
                               np.asarray(tvec, np.float64).reshape(3), float(K[0][0]), float(K[1][1]),
                               float(K[0][2]), float(K[1][2]))
        return out.astype(np.asarray(p3).dtype).reshape(-1, 1, 2), None

    cv_stub.getPerspectiveTransform = gpt
    cv_stub.fillPoly = fill_poly
    cv_stub.Rodrigues = rodrigues
    cv_stub.projectPoints = project_points
    cv_stub.DECOMP_SVD = 1


def gen_structure_oracle_patched():
    _patch_cv2_with_oracle()
    out = {}
    # (a) the reference's generate_remap_params loop on an MLS grid.
    for (h, w, seed, level) in [(96, 80, 1, 8), (64, 64, 0, 5)]:
        cfg = P_mls.SimilarityMlsConfigGenerator(P_mls.SimilarityMlsConfigGeneratorConfig(), level)((h, w), default_rng(seed))
        st = SimilarityMlsState(cfg, (h, w), None)
        map_y, map_x = st.src_image_grid.generate_remap_params(st.dst_image_grid)
        key = f'mls_{h}x{w}_s{seed}_l{level}'
        out[key + '_map_x'] = map_x
        out[key + '_map_y'] = map_y
        _, s_int = grid_to_arrays(st.src_image_grid)
        _, d_int = grid_to_arrays(st.dst_image_grid)
        out[key + '_src_grid'] = s_int
        out[key + '_dst_grid'] = d_int
    # (b) camera states (Rodrigues / projectPoints substituted).
    for name, gen_cls, cfg_cls, state_cls, shape, seed, level in [
        ('cubic', P_cam.CameraCubicCurveConfigGenerator, P_cam.CameraCubicCurveConfigGeneratorConfig,
         G_cam.CameraCubicCurveState, (96, 80), 0, 5),
        ('cubic', P_cam.CameraCubicCurveConfigGenerator, P_cam.CameraCubicCurveConfigGeneratorConfig,
         G_cam.CameraCubicCurveState, (200, 312), 3, 9),
        ('plane', P_cam.CameraPlaneOnlyConfigGenerator, P_cam.CameraPlaneOnlyConfigGeneratorConfig,
         G_cam.CameraPlaneOnlyState, (120, 90), 1, 7),
        ('fold', P_cam.CameraPlaneLineFoldConfigGenerator, P_cam.CameraPlaneLineFoldConfigGeneratorConfig,
         G_cam.CameraPlaneLineFoldState, (128, 160), 2, 6),
        ('curve', P_cam.CameraPlaneLineCurveConfigGenerator, P_cam.CameraPlaneLineCurveConfigGeneratorConfig,
         G_cam.CameraPlaneLineCurveState, (150, 110), 4, 3),
    ]:
        cfg = gen_cls(cfg_cls(), level)(shape, default_rng(seed))
        st = state_cls(cfg, shape, None)
        d_smooth, d_int = grid_to_arrays(st.dst_image_grid)
        key = f'cam_{name}_{shape[0]}x{shape[1]}_s{seed}_l{level}'
        out[key + '_dst_grid_smooth'] = d_smooth
        out[key + '_dst_grid'] = d_int
        out[key + '_result_shape'] = np.asarray(st.result_shape)
        out[key + '_config_json'] = np.frombuffer(json.dumps(plain(cfg)).encode(), dtype=np.uint8)
        # point projection through the grid (func_point), numpy + patched getPerspectiveTransform
        pt = Point.create(y=shape[0] * 0.37, x=shape[1] * 0.61)
        from vkit.mechanism.distortion.geometric.grid_rendering.interface import FuncImageGridBased
        q = FuncImageGridBased.func_point(cfg, st, shape, pt, None)
        out[key + '_point'] = np.asarray([pt.smooth_y, pt.smooth_x, q.smooth_y, q.smooth_x])
    np.savez_compressed(os.path.join(HERE, 'structure_oracle_patched.npz'), **out)


def gen_camera_states():
    """Camera states for the device-built path (vkx_camera_states_dev, csrc/camera.hip): the REFERENCE's own state constructors
    (CameraModel, the 2-D -> 3-D strategies, create_src_image_grid, create_dst_image_grid_and_shift_amounts_and_resize_ratios, Point
    rounding) on configs from its policy generators and on hand-made ones; cv.Rodrigues / cv.projectPoints are the oracle's
    restatements (cv2 is absent here), everything else -- numpy / OpenBLAS float32 and float64 arithmetic included -- runs for real.
    Stored: the config scalars as arrays, the destination lattice (rounded vertices), result shape and shift amounts."""
    _patch_cv2_with_oracle()
    out = {}
    cases = []
    rng = default_rng(2025)
    plan = [('cubic', (2048, 2048), 5), ('cubic', (2048, 2048), 9), ('plane', (2048, 2048), 7), ('cubic', (3072, 4096), 4)]
    for k in range(20):
        plan.append(('cubic' if k % 3 else 'plane', (int(rng.integers(40, 900)), int(rng.integers(40, 900))), int(rng.integers(1, 11))))
    for k, (kind, shape, level) in enumerate(plan):
        seed = 100 + k
        if kind == 'cubic':
            cfg = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), level)(shape, default_rng(seed))
        else:
            cfg = P_cam.CameraPlaneOnlyConfigGenerator(P_cam.CameraPlaneOnlyConfigGeneratorConfig(), level)(shape, default_rng(seed))
        if k % 5 == 4:      # explicit camera model fields instead of the completion from the shape
            cm = cfg.camera_model_config
            cm.principal_point = [float(shape[1]) * 0.4, float(shape[0]) * 0.55] + ([3.5] if k % 2 else [])
            cm.focal_length = float(max(shape)) * 1.3
            cm.camera_distance = float(max(shape)) * 0.9
        if k % 7 == 6:
            cfg.grid_size = 7   # a fine lattice: many pieces of the pairwise mean
        state_cls = G_cam.CameraCubicCurveState if kind == 'cubic' else G_cam.CameraPlaneOnlyState
        st = state_cls(cfg, shape, None)
        _, d_int = grid_to_arrays(st.dst_image_grid)
        cm = cfg.camera_model_config
        pp = list(cm.principal_point) if cm.principal_point else []
        cases.append([1.0 if kind == 'cubic' else 0.0, shape[0], shape[1], cfg.grid_size] + [float(v) for v in cm.rotation_unit_vec] +
                     [float(cm.rotation_theta), float(cm.focal_length or 0.0), float(cm.camera_distance or 0.0), float(len(pp))] +
                     [float(v) for v in (pp + [0.0, 0.0, 0.0])[:3]] +
                     ([float(cfg.curve_alpha), float(cfg.curve_beta), float(cfg.curve_direction), float(cfg.curve_scale)] if kind == 'cubic'
                      else [0.0, 0.0, 0.0, 0.0]))
        out[f'dst_{k}'] = np.asarray(d_int, np.int32)
        out[f'meta_{k}'] = np.asarray([st.result_shape[0], st.result_shape[1], st.shift_amount_y, st.shift_amount_x], np.int64)
    out['configs'] = np.asarray(cases, np.float64)
    out['columns'] = np.asarray(['cubic', 'height', 'width', 'grid_size', 'unit_x', 'unit_y', 'unit_z', 'theta', 'focal_length',
                                 'camera_distance', 'pp_len', 'pp0', 'pp1', 'pp2', 'curve_alpha', 'curve_beta', 'curve_direction', 'curve_scale'])
    np.savez_compressed(os.path.join(HERE, 'camera_states.npz'), **out)


if __name__ == '__main__':
    if len(sys.argv) > 1:                      # regenerate only the named fixtures: make_golden.py gen_mls_lattices ...
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    gen_numpy_path()
    gen_fill_modes()
    gen_pointwise_ops()
    gen_mls_states()
    gen_mls_lattices()
    gen_affine_states()
    gen_policy_configs()
    gen_operator_semantics()
    gen_random_distortion_sampling()
    gen_structure_oracle_patched()
    gen_page_resizing()
    gen_std_shift()
    gen_gcn()
    gen_camera_states()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
