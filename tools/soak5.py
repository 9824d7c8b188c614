#!/usr/bin/env python3
"""Fifth soak: the deterministic photometric operators (ellipse_streak included) end to end -- policy-sampled configs at random levels on random RGB
pages through ``DistortionPolicy.distort``, against the oracle called with the sampled config.
Usage: tools/soak5.py <seconds> <seed>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from numpy.random import default_rng

import oracle as O
from vkit_amd.element import Image
from vkit_amd.mechanism.distortion.photometric.opt import OutOfBoundBehavior
from vkit_amd.mechanism.distortion_policy.photometric import blur as P_blur, color as P_color, effect as P_effect, \
    streak as P_streak


def ksize(sigma):
    k = max(3, round(3 * sigma) + 1)
    return k + 1 if k % 2 == 0 else k


def std_shift(mat, cfg):
    sel = mat[:, :, list(cfg.channels)] if cfg.channels else mat
    f = sel.astype(np.float32)
    mean = np.mean(f.reshape(-1, f.shape[-1]), axis=0)
    f = f * cfg.scale - mean * (cfg.scale - 1)
    out = mat.copy()
    res = np.clip(np.round(f), 0, 255).astype(np.uint8)
    if cfg.channels:
        out[:, :, list(cfg.channels)] = res
    else:
        out = res
    return out


ORACLES = {
    'gaussian_blur': lambda m, c: O.gaussian_blur(m, ksize(c.sigma), c.sigma),
    'defocus_blur': lambda m, c: O.defocus_blur(m, c.radius, c.anti_aliasing_sigma),
    'motion_blur': lambda m, c: O.motion_blur(m, c.radius, c.angle, c.anti_aliasing_sigma),
    'zoom_in_blur': lambda m, c: O.zoom_in_blur(m, c.ratio, c.step, c.alpha),
    'mean_shift': lambda m, c: O.mean_shift(m, c.delta, c.threshold, c.channels, c.oob_behavior == OutOfBoundBehavior.CYCLE),
    'color_shift': lambda m, c: O.color_shift_rgb(m, c.delta),
    'brightness_shift': lambda m, c: O.brightness_shift_rgb(m, c.delta),
    'std_shift': std_shift,
    'boundary_equalization': lambda m, c: O.boundary_equalization(m, c.channels),
    'histogram_equalization': lambda m, c: O.histogram_equalization(m, c.channels),
    'complement': lambda m, c: O.complement(m, c.threshold, c.enable_threshold_lte, c.channels),
    'posterization': lambda m, c: O.posterization(m, c.num_bits, c.channels),
    'color_balance': lambda m, c: O.color_balance_rgb(m, c.ratio),
    'pixelation': lambda m, c: O.pixelation(m, c.ratio),
    'line_streak': lambda m, c: O.line_streak(m, c.thickness, c.gap, c.dash_thickness, c.dash_gap, c.color, c.alpha,
                                              c.enable_vert, c.enable_hori),
    'rectangle_streak': lambda m, c: O.rectangle_streak(m, c.thickness, c.aspect_ratio, c.dash_thickness, c.dash_gap,
                                                        c.short_side_min, c.short_side_step, c.color, c.alpha),
    'ellipse_streak': lambda m, c: O.ellipse_streak(m, c.thickness, c.aspect_ratio, c.short_side_min, c.short_side_step, c.color,
                                                    c.alpha),
}
factories = [P_blur.gaussian_blur_policy_factory, P_blur.defocus_blur_policy_factory, P_blur.motion_blur_policy_factory,
             P_blur.zoom_in_blur_policy_factory, P_color.mean_shift_policy_factory, P_color.color_shift_policy_factory,
             P_color.brightness_shift_policy_factory, P_color.std_shift_policy_factory,
             P_color.boundary_equalization_policy_factory, P_color.histogram_equalization_policy_factory,
             P_color.complement_policy_factory, P_color.posterization_policy_factory, P_color.color_balance_policy_factory,
             P_effect.pixelation_policy_factory, P_streak.line_streak_policy_factory,
             P_streak.rectangle_streak_policy_factory, P_streak.ellipse_streak_policy_factory]
policies = [f.create(None) for f in factories]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
counts = {}
while time.time() - t0 < budget:
    policy = policies[int(rng.integers(len(policies)))]
    h, w = int(rng.integers(24, 300)), int(rng.integers(24, 300))
    level = int(rng.integers(1, 11))
    mat = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if rng.random() < 0.15:
        mat = (mat // 64 * 64).astype(np.uint8)      # few grey levels: flat channels, histogram corner cases
    seed = int(rng.integers(1 << 30))
    res = policy.distort(level, image=Image(mat=mat), rng=default_rng(seed), enable_debug=True)
    want = ORACLES[policy.name](mat, res.config)
    assert res.image.mat.shape == want.shape and (res.image.mat == want).all(), (policy.name, level, (h, w), seed, res.config)
    counts[policy.name] = counts.get(policy.name, 0) + 1
print('soak5 ok', counts, round(time.time() - t0), 's')
