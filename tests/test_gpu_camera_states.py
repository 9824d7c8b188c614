"""Device-built camera states (``vkx_camera_states_dev``, csrc/camera.hip) through the C ABI: the vertex lattices of a batch of
configs against the lattices the REFERENCE's own state constructors produced (tests/golden/camera_states.npz), and
``ChainBatch.add_config`` -- state construction inside ``run`` -- against ``ChainBatch.add`` on host-built states and against the
oracle chain."""
import ctypes

import numpy as np
import pytest
from numpy.random import default_rng

import oracle as O
from test_camera_states import golden_cases

pytestmark = pytest.mark.gpu


def _build(N, ctx, cases, stream):
    n = len(cases)
    records = (N.VkxCameraConfig * n)(*[N.camera_config(cfg, shape) for cfg, shape, _w, _m in cases])
    lattices = []
    for k in range(n):
        model = N.VkxCameraModel()
        N.check(N.lib().vkx_camera_model_host(ctypes.byref(records[k]), ctypes.byref(model)))
        lattices.append((ctx.dev_empty((model.rows, model.cols, 2), np.int32), ctx.dev_empty((model.rows, model.cols, 2), np.int32)))
    sv = (ctypes.c_void_p * n)(*[a.ptr for a, _b in lattices])
    dv = (ctypes.c_void_p * n)(*[b.ptr for _a, b in lattices])
    out_ptr = ctx.host_alloc(n * ctypes.sizeof(N.VkxGridState))
    out = (N.VkxGridState * n).from_address(out_ptr)
    N.check(N.lib().vkx_camera_states_dev(ctx.handle, records, n, sv, dv, ctypes.c_void_p(out_ptr), stream))
    ctx.sync_stream(stream)
    states = [(out[k].rows, out[k].cols, out[k].dh, out[k].dw, out[k].shift_y, out[k].shift_x, out[k].flags) for k in range(n)]
    ctx.host_free(out_ptr)
    return lattices, states


@pytest.mark.parametrize('stream', [0, 2])
def test_lattices_equal_the_reference_goldens(golden_dir, stream):
    from vkit_amd import _native as N
    from vkit_amd.mechanism.distortion.geometric.grid_rendering.grid_creator import create_src_image_grid
    ctx = N.default_ctx()
    cases = list(golden_cases(golden_dir))
    lattices, states = _build(N, ctx, cases, stream)
    for k, ((cfg, shape, want, (dh, dw, sy, sx)), (sv, dv), st) in enumerate(zip(cases, lattices, states)):
        assert st[6] == 0, k
        assert st[:6] == (want.shape[0], want.shape[1], dh, dw, sy, sx), (k, st)
        assert np.array_equal(dv.host(), want), k
        assert np.array_equal(sv.host(), create_src_image_grid(shape[0], shape[1], cfg.grid_size).vertices), k


def test_lattices_equal_the_host_operator_on_policy_configs():
    """300 configs of the policy generators at random levels and page sizes in ONE call: every lattice, shape and shift is the host
    operator's (numpy + OpenBLAS of this box: the float32 accumulation orders the C half restates are those of the box the goldens
    were generated on -- a box whose OpenBLAS picks other kernels fails HERE first, and only here)."""
    from vkit_amd import _native as N
    from vkit_amd.mechanism import distortion as D
    from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam
    ctx = N.default_ctx()
    rng = default_rng(77)
    cases, hosts = [], []
    for k in range(300):
        shape = (int(rng.integers(30, 1200)), int(rng.integers(30, 1200)))
        level = int(rng.integers(1, 11))
        if k % 4:
            cfg = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), level)(shape, rng)
            st = D.camera_cubic_curve.generate_state(cfg, shape)
        else:
            cfg = P_cam.CameraPlaneOnlyConfigGenerator(P_cam.CameraPlaneOnlyConfigGeneratorConfig(), level)(shape, rng)
            st = D.camera_plane_only.generate_state(cfg, shape)
        cases.append((cfg, shape, None, None))
        hosts.append(st)
    lattices, states = _build(N, ctx, cases, 0)
    differing = []
    for k, (st, (sv, dv), got) in enumerate(zip(hosts, lattices, states)):
        same = (got[6] == 0 and got[2:6] == (st.result_shape[0], st.result_shape[1], st.shift_amount_y, st.shift_amount_x)
                and np.array_equal(dv.host(), st.dst_image_grid.vertices) and np.array_equal(sv.host(), st.src_image_grid.vertices))
        if not same:
            differing.append(k)
    assert not differing, differing


def test_non_finite_projection_is_reported():
    """A focal length beyond float32 (the intrinsic matrix is float32: inf) projects the vertices to inf and 0 x inf = NaN: the reference
    fails in Point's round() with ValueError; the device flags the state and ``ChainBatch.run`` raises the same exception."""
    from vkit_amd import _native as N
    from vkit_amd.batch import ChainBatch
    from vkit_amd.mechanism.distortion.geometric import camera as C
    cm = C.CameraModelConfig(rotation_unit_vec=[1.0, 0.0, 0.0], rotation_theta=0.0, focal_length=1e39, camera_distance=100.0,
                             principal_point=[0.0, 0.0])
    cfg = C.CameraPlaneOnlyConfig(camera_model_config=cm, grid_size=16)
    _lat, states = _build(N, N.default_ctx(), [(cfg, (64, 64), None, None)], 0)
    assert states[0][6] != 0
    batch = ChainBatch()
    batch.add_config(np.zeros((64, 64, 3), np.uint8), cfg)
    with pytest.raises(ValueError):
        batch.run()
    batch.close()
    from vkit_amd.mechanism import distortion as D
    with pytest.raises(ValueError), np.errstate(all='ignore'):
        D.camera_plane_only.generate_state(cfg, (64, 64))


def test_chain_batch_builds_its_states_inside_run():
    """``add_config`` against ``add``: same pixels from configs as from host-built states -- ragged shapes, both camera kinds, with
    and without noise streams and photometric members, three runs (the lattice sets alternate), a replaced config."""
    from vkit_amd.batch import ChainBatch
    from vkit_amd.mechanism import distortion as D
    from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam
    rng = default_rng(5)
    items = []
    for k in range(22):
        shape = (int(rng.integers(60, 330)), int(rng.integers(60, 330)))
        level = int(rng.integers(1, 11))
        if k % 3:
            cfg = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), level)(shape, rng)
            op = D.camera_cubic_curve
        else:
            cfg = P_cam.CameraPlaneOnlyConfigGenerator(P_cam.CameraPlaneOnlyConfigGeneratorConfig(), level)(shape, rng)
            op = D.camera_plane_only
        image = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
        kwargs = dict(blur_sigma=1.0 if k % 2 else None, hue_delta=37 if k % 4 else None)
        noisy = k % 5 != 0
        items.append((image, cfg, op, kwargs, noisy))

    def run(from_config):
        batch = ChainBatch()
        for k, (image, cfg, op, kwargs, noisy) in enumerate(items):
            noise = dict(noise_std=9.0, noise_rng=default_rng(800 + k)) if noisy else {}
            if from_config:
                batch.add_config(image, cfg, **kwargs, **noise)
            else:
                batch.add(image, op.generate_state(cfg, image.shape[:2]), **kwargs, **noise)
        for _ in range(3):
            batch.run()
        out = [batch.result(k) for k in range(len(items))]
        return batch, out

    want_batch, want = run(False)
    got_batch, got = run(True)
    assert got_batch.stream_fallbacks == 0
    for k, (g, w) in enumerate(zip(got, want)):
        assert g.shape == w.shape and (g == w).all(), k
    # one image against the oracle chain fed numpy's own plane
    image, cfg, op, kwargs, _noisy = items[1]
    st = op.generate_state(cfg, image.shape[:2])
    mx, my = O.grid_to_map(st.src_image_grid.vertices, st.dst_image_grid.vertices, st.result_shape)
    ref = O.remap(image, mx, my)
    if kwargs['blur_sigma']:
        ref = O.gaussian_blur(ref, 5, 1.0)
    if kwargs['hue_delta']:
        ref = O.color_shift_rgb(ref, 37)
    plane = np.round(default_rng(801).normal(0, 9.0, tuple(st.result_shape) + (3,))).astype(np.int16)
    assert (got[1] == O.add_noise_i16(ref, plane)).all()
    # a replaced config: new shape, new buffers
    image, _cfg, _op, kwargs, _n = items[2]
    new_cfg = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), 9)(image.shape[:2], default_rng(4242))
    got_batch.set_config(2, new_cfg)
    got_batch.run()
    st = D.camera_cubic_curve.generate_state(new_cfg, image.shape[:2])
    single = ChainBatch()
    single.add(image, st, **kwargs, noise_std=9.0, noise_rng=default_rng(802))
    single.run()
    assert (got_batch.result(2) == single.result(0)).all()
    assert (got_batch.result(7) == want[7]).all()
    for b in (want_batch, got_batch, single):
        b.close()
