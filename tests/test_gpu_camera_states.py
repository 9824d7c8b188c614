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


# ---------------------------------------------------------------------------------------------- similarity_mls, batched
def _build_mls(N, ctx, cases, stream=0):
    """cases: (SimilarityMlsConfig, shape); returns (lattice DevArrays, result tuples) of ONE vkx_mls_states_dev call."""
    n = len(cases)
    recs, keep = zip(*[N.mls_config(cfg, shape) for cfg, shape in cases])
    records = (N.VkxMlsConfig * n)(*recs)
    lattices = []
    for rec in recs:
        rows, cols = N.lattice_shape(rec.height, rec.width, rec.grid_size)
        lattices.append((ctx.dev_empty((rows, cols, 2), np.int32), ctx.dev_empty((rows, cols, 2), np.int32)))
    sv = (ctypes.c_void_p * n)(*[a.ptr for a, _b in lattices])
    dv = (ctypes.c_void_p * n)(*[b.ptr for _a, b in lattices])
    out_ptr = ctx.host_alloc(n * ctypes.sizeof(N.VkxGridState))
    out = (N.VkxGridState * n).from_address(out_ptr)
    N.check(N.lib().vkx_mls_states_dev(ctx.handle, records, n, sv, dv, ctypes.c_void_p(out_ptr), stream))
    ctx.sync_stream(stream)
    states = [(out[k].rows, out[k].cols, out[k].dh, out[k].dw, out[k].shift_y, out[k].shift_x, out[k].flags) for k in range(n)]
    ctx.host_free(out_ptr)
    del keep
    return lattices, states


def test_mls_states_batch_equals_the_reference_goldens(golden_dir):
    """All twelve similarity_mls goldens of the imported reference (five small states, 1024^2, the C2 / C5 sizes 2048^2 and 4096^2) in
    ONE vkx_mls_states_dev call: lattices, result shapes and shifts bit for bit."""
    import json
    import os
    from vkit_amd import _native as N
    from vkit_amd.element import Point, PointTuple
    from vkit_amd.mechanism import distortion as D
    cases, wants = [], []
    for fname in ('mls_states.npz', 'mls_lattices.npz'):
        M = np.load(os.path.join(golden_dir, fname))
        for m in json.loads(bytes(M['meta_json'])):
            k = m['key']
            if k + '_src_handles' not in M.files:
                continue
            cfg = D.SimilarityMlsConfig(
                src_handle_points=PointTuple(Point.create(y=y, x=x) for x, y in M[k + '_src_handles']),
                dst_handle_points=PointTuple(Point.create(y=y, x=x) for x, y in M[k + '_dst_handles']),
                grid_size=m['grid_size'])
            cases.append((cfg, (m['h'], m['w'])))
            wants.append((np.asarray(M[k + '_dst_grid'], np.int32), tuple(m['result_shape']), m['shift']))
    assert len(cases) == 12
    for stream in (0, 2):
        lattices, states = _build_mls(N, N.default_ctx(), cases, stream)
        for k, ((_sv, dv), st, (want, shape, shift)) in enumerate(zip(lattices, states, wants)):
            assert st[6] == 0 and (st[2], st[3]) == shape and [st[4], st[5]] == shift, (k, st)
            assert np.array_equal(dv.host(), want), k


def test_mls_states_batch_equals_the_host_operator(monkeypatch):
    """Policy configs at random levels on random page shapes -- elongated pages included (hundreds of handles: the pairwise sums and the
    sgemv kernel switch) -- in one call, against the operator's own state (which projects lattice by lattice)."""
    from vkit_amd import _native as N
    from vkit_amd.mechanism import distortion as D
    from vkit_amd.mechanism.distortion_policy.geometric import mls as P_mls
    monkeypatch.delenv('VKX_MLS_HOST_PROJECTION', raising=False)
    rng = default_rng(31)
    cases, hosts = [], []
    for k in range(60):
        shape = (int(rng.integers(60, 900)), int(rng.integers(60, 900))) if k % 6 else (int(rng.integers(1500, 2600)), int(rng.integers(90, 160)))
        cfg = P_mls.SimilarityMlsConfigGenerator(P_mls.SimilarityMlsConfigGeneratorConfig(), int(rng.integers(1, 11)))(shape, rng)
        cases.append((cfg, shape))
        hosts.append(D.similarity_mls.generate_state(cfg, shape))
    lattices, states = _build_mls(N, N.default_ctx(), cases)
    for k, (st, (sv, dv), got) in enumerate(zip(hosts, lattices, states)):
        assert got[6] == 0 and got[2:6] == (st.result_shape[0], st.result_shape[1], st.shift_amount_y, st.shift_amount_x), (k, got)
        assert np.array_equal(dv.host(), st.dst_image_grid.vertices) and np.array_equal(sv.host(), st.src_image_grid.vertices), k


def test_chain_batch_mixes_camera_and_mls_configs():
    """``add_config`` with similarity_mls and camera configs in one batch (two state calls, one layout) against ``add`` on host-built
    states; a vertex ON an integer handle position raises FloatingPointError like the reference's np.errstate(divide='raise')."""
    from vkit_amd.batch import ChainBatch
    from vkit_amd.element import Point, PointTuple
    from vkit_amd.mechanism import distortion as D
    from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam, mls as P_mls
    rng = default_rng(8)
    items = []
    for k in range(18):
        shape = (int(rng.integers(70, 300)), int(rng.integers(70, 300)))
        level = int(rng.integers(1, 11))
        if k % 2:
            cfg, op = P_mls.SimilarityMlsConfigGenerator(P_mls.SimilarityMlsConfigGeneratorConfig(), level)(shape, rng), D.similarity_mls
        else:
            cfg, op = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), level)(shape, rng), D.camera_cubic_curve
        items.append((rng.integers(0, 256, shape + (3,), dtype=np.uint8), cfg, op))

    def run(from_config):
        batch = ChainBatch()
        for k, (image, cfg, op) in enumerate(items):
            kwargs = dict(blur_sigma=1.0, hue_delta=21, noise_std=7.0, noise_rng=default_rng(300 + k))
            if from_config:
                batch.add_config(image, cfg, **kwargs)
            else:
                batch.add(image, op.generate_state(cfg, image.shape[:2]), **kwargs)
        batch.run()
        batch.run()
        out = [batch.result(k) for k in range(len(items))]
        batch.close()
        return out

    for k, (g, w) in enumerate(zip(run(True), run(False))):
        assert g.shape == w.shape and (g == w).all(), k
    # a lattice vertex (0, 0) on an integer handle position whose smooth position differs: 1 / 0 in the weights
    handles = PointTuple([Point.create(y=0.2, x=0.3), Point.create(y=40, x=50), Point.create(y=10, x=60)])
    cfg = D.SimilarityMlsConfig(src_handle_points=handles, dst_handle_points=handles, grid_size=16)
    batch = ChainBatch()
    batch.add_config(np.zeros((64, 80, 3), np.uint8), cfg)
    with pytest.raises(FloatingPointError):
        batch.run()
    batch.close()
    with pytest.raises(FloatingPointError):
        D.similarity_mls.generate_state(cfg, (64, 80))
