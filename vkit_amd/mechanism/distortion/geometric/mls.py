"""``similarity_mls``: moving-least-squares similarity deformation of the vertex lattice (reference:
vkit/mechanism/distortion/geometric/mls.py; Schaefer, McPhail, Warren, "Image deformation using moving least
squares", 2006, section 2.2).

Everything downstream only sees the ROUNDED vertex, so a last-bit difference in the float32 arithmetic can flip a
rounding: the lattice is projected with the reference's float32 operations in the reference's order, including the
accumulation orders of numpy's reductions and of the BLAS kernels behind ``np.matmul``.  The whole lattice goes
through ONE device launch (``vkx_mls_project``, csrc/mls.hip: 10 816 vertices x 25 handles at 2048^2 in a few
microseconds instead of 0.3 - 0.8 s of per-vertex numpy calls); ``project_point`` keeps the reference's
single-vertex numpy form.  ``VKX_MLS_HOST_PROJECTION=1`` routes the lattice through ``project_point`` vertex by
vertex -- the reference's own path, for building states on a host without a GPU (the CPU tests); it is never
selected implicitly.  Device side as well: the dense remap through the two integer lattices (grid_rendering).
"""
import os
from typing import Optional, Tuple

import attrs
import numpy as np
from numpy.random import Generator as RandomGenerator

from vkit_amd import _native
from vkit_amd.element import Point, PointTuple
from ..interface import DistortionConfig
from .grid_rendering.grid_creator import create_src_image_grid
from .grid_rendering.interface import DistortionImageGridBased, DistortionStateImageGridBased
from .grid_rendering.point_projector import PointProjector


@attrs.define
class SimilarityMlsConfig(DistortionConfig):
    src_handle_points: PointTuple
    dst_handle_points: PointTuple
    grid_size: int
    resize_as_src: bool = False


class SimilarityMlsPointProjector(PointProjector):

    def __init__(self, src_handle_points: PointTuple, dst_handle_points: PointTuple):
        self.src_handle_points = src_handle_points
        self.dst_handle_points = dst_handle_points
        # a vertex sitting exactly on a handle maps to that handle's target (weights would be infinite)
        self.pinned = {(p.smooth_x, p.smooth_y): q for p, q in zip(src_handle_points, dst_handle_points)}
        # (N, 2) float32, integer handle positions (PointTuple.to_smooth_np_array quirk)
        self.p = src_handle_points.to_smooth_np_array()
        self.q = dst_handle_points.to_smooth_np_array()

    def project_point(self, src_point: Point):
        vx, vy = src_point.smooth_x, src_point.smooth_y
        hit = self.pinned.get((vx, vy))
        if hit is not None:
            return hit

        p, q = self.p, self.q
        # w_i = 1 / |p_i - v|^2
        delta = p.copy()
        delta[:, 0] -= vx
        delta[:, 1] -= vy
        np.square(delta, out=delta)
        dist2 = np.sum(delta, axis=1)
        with np.errstate(divide='raise'):
            w = 1 / dist2
            w_norm = w / np.sum(w)

        # weighted centroids p*, q* and the centred handles
        p_star = np.matmul(w_norm, p)
        q_star = np.matmul(w_norm, q)
        p_hat = p - p_star
        q_hat = q - q_star
        p_hat_perp = p_hat[:, [1, 0]]
        p_hat_perp[:, 0] *= -1

        # A_i = w_i * [p_hat_i ; -p_hat_i^perp] [v - p* ; -(v - p*)^perp]^T
        px, py = p_star
        anchor = np.transpose(np.asarray([(vx - px, vy - py), (vy - py, -(vx - px))], dtype=np.float32))
        top = np.matmul(p_hat, anchor)
        bottom = np.matmul(-p_hat_perp, anchor)
        A = np.expand_dims(np.expand_dims(w, axis=1), axis=1) * np.stack((top, bottom), axis=1)

        # f(v) = sum_i q_hat_i A_i / mu_s + q*
        terms = np.squeeze(np.matmul(np.expand_dims(q_hat, axis=1), A), axis=1)
        mu = np.sum(w * np.sum(p_hat * p_hat, axis=1))
        fx, fy = np.sum(terms, axis=0) / mu + q_star
        return Point.create(y=float(fy), x=float(fx))

    def project_array(self, smooth_xy: np.ndarray) -> np.ndarray:
        if os.environ.get('VKX_MLS_HOST_PROJECTION', '') == '1':
            return super().project_array(smooth_xy)
        smooth = lambda pts: np.asarray([(p.smooth_x, p.smooth_y) for p in pts], dtype=np.float64)   # noqa: E731
        return _native.mls_project(self.p, self.q, smooth(self.src_handle_points), smooth(self.dst_handle_points),
                                   smooth_xy)


class SimilarityMlsState(DistortionStateImageGridBased[SimilarityMlsConfig]):

    def __init__(self, config: SimilarityMlsConfig, shape: Tuple[int, int], rng: Optional[RandomGenerator]):
        height, width = shape
        self.initialize_image_grid_based(
            create_src_image_grid(height, width, config.grid_size),
            SimilarityMlsPointProjector(config.src_handle_points, config.dst_handle_points),
            resize_as_src=config.resize_as_src,
        )
        # debug aid, as in the reference
        self.dst_handle_points = list(map(self.shift_and_resize_point, config.dst_handle_points))


similarity_mls = DistortionImageGridBased(config_cls=SimilarityMlsConfig, state_cls=SimilarityMlsState)
