#!/usr/bin/env python3
"""Fourth soak: the ten geometric operators end to end -- policy-sampled configs at random levels on random page sizes,
Image + Mask + ScoreMap (+ active mask, points) through ``DistortionPolicy.distort``, against the oracle evaluated on the
state the operator reports (lattice -> dense map -> remap, or the affine matrix -> warp).
Usage: tools/soak4.py <seconds> <seed>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from numpy.random import default_rng

import oracle as O
from vkit_amd.element import Image, Mask, Point, PointList, ScoreMap
from vkit_amd.mechanism.distortion_policy.geometric import affine as P_aff, camera as P_cam, mls as P_mls

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
factories = [P_aff.shear_hori_policy_factory, P_aff.shear_vert_policy_factory, P_aff.rotate_policy_factory,
             P_aff.skew_hori_policy_factory, P_aff.skew_vert_policy_factory, P_mls.similarity_mls_policy_factory,
             P_cam.camera_plane_only_policy_factory, P_cam.camera_cubic_curve_policy_factory,
             P_cam.camera_plane_line_fold_policy_factory, P_cam.camera_plane_line_curve_policy_factory]
policies = [f.create(None) for f in factories]
t0 = time.time()
counts = {}
while time.time() - t0 < budget:
    policy = policies[int(rng.integers(len(policies)))]
    h, w = int(rng.integers(40, 420)), int(rng.integers(40, 420))
    level = int(rng.integers(1, 11))
    image = Image(mat=rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    mask = Mask(mat=(rng.random((h, w)) < 0.5).astype(np.uint8))
    score = ScoreMap(mat=rng.random((h, w), dtype=np.float32))
    pts = PointList(Point.create(y=int(y), x=int(x)) for y, x in zip(rng.integers(0, h - 1, 6), rng.integers(0, w - 1, 6)))   # the last row / column can fall outside the cell table, as in the reference
    seed = int(rng.integers(1 << 30))
    res = policy.distort(level, image=image, mask=mask, score_map=score, points=pts, rng=default_rng(seed),
                         enable_debug=True)
    state = res.state
    if state is None or res.image is image:        # a no-op draw (angle 0 and the like)
        counts['noop'] = counts.get('noop', 0) + 1
        continue
    if hasattr(state, 'src_image_grid'):
        sv, dv = state.src_image_grid.vertices, state.dst_image_grid.vertices
        shape = state.result_shape
        mx, my = O.grid_to_map(sv, dv, shape)
        want = [O.remap(image.mat, mx, my), O.remap(mask.mat, mx, my), O.remap(score.mat, mx, my)]
    else:
        M, dsize = np.asarray(state.trans_mat, np.float64), state.dsize
        warp = O.warp_affine if M.shape[0] == 2 else O.warp_perspective
        want = [warp(image.mat, M, dsize), warp(mask.mat, M, dsize), warp(score.mat, M, dsize)]
    for got, exp, what in zip((res.image.mat, res.mask.mat, res.score_map.mat), want, ('image', 'mask', 'score_map')):
        assert got.shape == exp.shape and (got == exp).all(), (policy.name, level, (h, w), seed, what)
    # (PointTuple.from_np_array drops a closing duplicate -- first == last -- like the reference, element/point.py:156-166:
    #  the affine family returns 5 points when the first and the last random point coincide after the transform)
    assert len(res.points) in (5, 6), (policy.name, level, (h, w), seed, len(res.points), [p.to_xy_pair() for p in pts])
    counts[policy.name] = counts.get(policy.name, 0) + 1
print('soak4 ok', counts, round(time.time() - t0), 's')
