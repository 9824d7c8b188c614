"""The host half of the device-built camera states (``vkx_camera_model_host``, csrc/camera.hip) against the Python operators it
restates (vkit_amd/mechanism/distortion/geometric/camera.py = reference camera.py:58-265, 324-423): Rodrigues matrix, float32
translation, intrinsics, the cubic curve's direction row / extent / polynomial, the lattice shape -- bit for bit, on configs drawn by
the reference-compatible policy generators and on hand-made ones (explicit principal point / focal length, degenerate rotations).
No GPU: the function is pure host C.  (The float32 accumulation orders are those of numpy + OpenBLAS on the host the goldens come
from; a host whose OpenBLAS picks other kernels rounds its own CameraModel differently -- DESIGN section 2.)"""
import numpy as np
import pytest
from numpy.random import default_rng

from vkit_amd import _native as N
from vkit_amd.mechanism.distortion.geometric import camera as C
from vkit_amd.mechanism.distortion.geometric.grid_rendering.grid_creator import create_src_image_grid
from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam


def _check(config, shape):
    h, w = shape
    got = N.camera_model_host(config, shape)
    cm = C.CameraModel(C.DistortionStateCameraOperation.complete_camera_model_config(h, w, config.camera_model_config))
    R = C.rodrigues(np.asarray(cm.rotation_vec, dtype=np.float64))
    assert np.array_equal(np.asarray(got.R[:]).reshape(3, 3), R)
    assert np.array_equal(np.asarray(got.t[:]), np.asarray(cm.translation_vec, np.float64).reshape(3))
    K = np.asarray(cm.intrinsic_mat, np.float64)
    assert (got.fx, got.fy, got.cx, got.cy) == (K[0, 0], K[1, 1], K[0, 2], K[1, 2])
    grid = create_src_image_grid(h, w, config.grid_size)
    assert (got.rows, got.cols) == grid.shape
    if hasattr(config, 'curve_scale'):
        st = C.CameraCubicCurvePoint2dTo3dStrategy(h, w, config.curve_alpha, config.curve_beta, config.curve_direction, config.curve_scale)
        assert np.float32(got.a0) == st.rotation_mat[0, 0] and np.float32(got.a1) == st.rotation_mat[0, 1]
        assert np.float32(got.along_min) == st.plane_projection_min and np.float32(got.along_range) == st.plane_projection_range
        poly = [st.curve_alpha + st.curve_beta, -2 * st.curve_alpha - st.curve_beta, st.curve_alpha, 0.0]
        assert list(got.poly[:]) == poly and got.curve_scale == st.curve_scale and got.points_f32 == 0
    else:
        assert got.points_f32 == 1


@pytest.mark.parametrize('seed', range(8))
def test_host_scalars_equal_the_python_operators_on_policy_configs(seed):
    rng = default_rng(seed)
    for _ in range(150):
        shape = (int(rng.integers(16, 2300)), int(rng.integers(16, 2300)))
        level = int(rng.integers(1, 11))
        if rng.random() < 0.7:
            gen = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), level)
        else:
            gen = P_cam.CameraPlaneOnlyConfigGenerator(P_cam.CameraPlaneOnlyConfigGeneratorConfig(), level)
        _check(gen(shape, rng), shape)


def test_host_scalars_on_hand_made_configs():
    rng = default_rng(99)
    for k in range(300):
        shape = (int(rng.integers(8, 900)), int(rng.integers(8, 900)))
        vec = rng.normal(0, 1, 3)
        if k % 7 == 0:
            vec = np.array([0.0, 0.0, 1.0])            # already a unit vector: no division
        if k % 11 == 0:
            theta = 0.0                                # identity rotation
        else:
            theta = float(rng.uniform(-120, 120))      # clipped to +-89
        cm = C.CameraModelConfig(rotation_unit_vec=[float(v) for v in vec], rotation_theta=theta)
        if k % 3 == 0:
            cm.principal_point = [float(rng.uniform(0, shape[1])), float(rng.uniform(0, shape[0]))] + ([float(rng.uniform(-5, 5))] if k % 2 else [])
            cm.focal_length = float(rng.uniform(50, 3000))
            cm.camera_distance = float(rng.uniform(50, 3000))
        elif k % 3 == 1:
            cm.focal_length = float(rng.uniform(50, 3000))     # camera_distance missing: both are completed from the shape
        grid_size = int(rng.integers(3, 80))
        if k % 2:
            cfg = C.CameraCubicCurveConfig(curve_alpha=float(rng.uniform(-100, 100)), curve_beta=float(rng.uniform(-100, 100)),
                                           curve_direction=float(rng.uniform(-400, 400)), curve_scale=float(rng.uniform(0.1, 2.0)),
                                           camera_model_config=cm, grid_size=grid_size)
        else:
            cfg = C.CameraPlaneOnlyConfig(camera_model_config=cm, grid_size=grid_size)
        _check(cfg, shape)


def golden_cases(golden_dir):
    """(config object, shape, expected dst lattice, (dh, dw, shift_y, shift_x)) of tests/golden/camera_states.npz -- lattices the
    REFERENCE's own state constructors produced (tests/golden/make_golden.py gen_camera_states)."""
    import os
    data = np.load(os.path.join(golden_dir, 'camera_states.npz'))
    cols = [str(c) for c in data['columns']]
    for k, row in enumerate(data['configs']):
        f = dict(zip(cols, row))
        cm = C.CameraModelConfig(rotation_unit_vec=[f['unit_x'], f['unit_y'], f['unit_z']], rotation_theta=f['theta'])
        if f['pp_len']:
            cm.principal_point = [f['pp0'], f['pp1'], f['pp2']][:int(f['pp_len'])]
        if f['focal_length']:
            cm.focal_length = f['focal_length']
        if f['camera_distance']:
            cm.camera_distance = f['camera_distance']
        if f['cubic']:
            cfg = C.CameraCubicCurveConfig(curve_alpha=f['curve_alpha'], curve_beta=f['curve_beta'], curve_direction=f['curve_direction'],
                                           curve_scale=f['curve_scale'], camera_model_config=cm, grid_size=int(f['grid_size']))
        else:
            cfg = C.CameraPlaneOnlyConfig(camera_model_config=cm, grid_size=int(f['grid_size']))
        yield cfg, (int(f['height']), int(f['width'])), data[f'dst_{k}'], tuple(int(v) for v in data[f'meta_{k}'])


def test_host_states_equal_the_reference_goldens(golden_dir):
    """The host operators of this package on the golden configs: the lattices the device path is held to are the reference's."""
    from vkit_amd.mechanism import distortion as D
    n = 0
    for cfg, shape, want, (dh, dw, sy, sx) in golden_cases(golden_dir):
        op = D.camera_cubic_curve if hasattr(cfg, 'curve_scale') else D.camera_plane_only
        st = op.generate_state(cfg, shape)
        assert np.array_equal(st.dst_image_grid.vertices, want), n
        assert tuple(st.result_shape) == (dh, dw) and (st.shift_amount_y, st.shift_amount_x) == (sy, sx)
        _check(cfg, shape)
        n += 1
    assert n >= 20


def test_tile_buffer_layout_formula():
    """ChainBatch lays the tile buffers of a batch out with whole-array numpy arithmetic: the formula is vkx_np_tiles_layout's."""
    import ctypes
    rng = default_rng(4)
    for n in [1, 2, 3071, 3072, 3073, 98304, 14_500_000, 0x7fffffff] + [int(v) for v in rng.integers(1, 60_000_000, 200)]:
        vals = [ctypes.c_int64() for _ in range(5)]
        N.check(N.lib().vkx_np_tiles_layout(n, *[ctypes.byref(v) for v in vals]))
        tiles = (n + n // 45 + 4096 + 3072 - 1) // 3072
        slots_off = (16 + 8 * (tiles + 1) + 255) & ~255
        assert (vals[0].value, vals[2].value, vals[3].value, vals[4].value) == (tiles, 16, slots_off, slots_off + tiles * vals[1].value * 2), n
