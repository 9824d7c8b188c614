"""Host-side logic of vkit_amd (operator, states, policies) against golden vectors produced by the real reference.
CPU only: nothing here launches a kernel."""
import json
import os

import attrs
import numpy as np
import pytest

from vkit_amd.mechanism.distortion.interface import DistortionResult
from numpy.random import default_rng

from vkit_amd.element import Box, Image, ImageMode, Mask, Point, PointList, PointTuple, Polygon, ScoreMap
from vkit_amd.mechanism import distortion as D
from vkit_amd.mechanism.distortion.geometric.grid_rendering.interface import FuncImageGridBased
from vkit_amd.mechanism.distortion.interface import Distortion, DistortionConfig, DistortionNopState
from vkit_amd.mechanism.distortion_policy import RandomDistortionFactoryConfig, random_distortion_factory
from vkit_amd.mechanism.distortion_policy.geometric import affine as P_aff, camera as P_cam, mls as P_mls
from vkit_amd.mechanism.distortion_policy.photometric import blur as P_blur, color as P_color, effect as P_effect, \
    noise as P_noise, streak as P_streak
from vkit_amd.utility import dyn_structure


@pytest.fixture(autouse=True)
def _mls_lattice_on_the_host(monkeypatch):
    """No GPU here: similarity_mls states are built through the reference's per-vertex ``project_point`` (the device
    path, ``vkx_mls_project``, is checked against the same goldens in tests/test_gpu_operators.py)."""
    monkeypatch.setenv('VKX_MLS_HOST_PROJECTION', '1')


def plain(obj):
    if isinstance(obj, Point):
        return [obj.smooth_y, obj.smooth_x]
    if attrs.has(type(obj)):
        return {a.name.lstrip('_'): plain(getattr(obj, a.name)) for a in attrs.fields(type(obj)) if a.name != '_rng_state'}
    if isinstance(obj, (list, tuple)):
        return [plain(v) for v in obj]
    if isinstance(obj, np.integer):
        return int(obj)
    if isinstance(obj, np.floating):
        return float(obj)
    if hasattr(obj, 'value') and not isinstance(obj, (int, float, str)):
        return obj.value
    return obj


GENERATORS = {
    'similarity_mls': (P_mls.SimilarityMlsConfigGenerator, P_mls.SimilarityMlsConfigGeneratorConfig),
    'camera_plane_only': (P_cam.CameraPlaneOnlyConfigGenerator, P_cam.CameraPlaneOnlyConfigGeneratorConfig),
    'camera_cubic_curve': (P_cam.CameraCubicCurveConfigGenerator, P_cam.CameraCubicCurveConfigGeneratorConfig),
    'camera_plane_line_fold': (P_cam.CameraPlaneLineFoldConfigGenerator, P_cam.CameraPlaneLineFoldConfigGeneratorConfig),
    'camera_plane_line_curve': (P_cam.CameraPlaneLineCurveConfigGenerator, P_cam.CameraPlaneLineCurveConfigGeneratorConfig),
    'shear_hori': (P_aff.ShearHoriConfigGenerator, P_aff.ShearHoriConfigGeneratorConfig),
    'std_shift': (P_color.StdShiftConfigGenerator, P_color.StdShiftConfigGeneratorConfig),
    'defocus_blur': (P_blur.DefocusBlurConfigGenerator, P_blur.DefocusBlurConfigGeneratorConfig),
    'motion_blur': (P_blur.MotionBlurConfigGenerator, P_blur.MotionBlurConfigGeneratorConfig),
    'shear_vert': (P_aff.ShearVertConfigGenerator, P_aff.ShearVertConfigGeneratorConfig),
    'rotate': (P_aff.RotateConfigGenerator, P_aff.RotateConfigGeneratorConfig),
    'skew_hori': (P_aff.SkewHoriConfigGenerator, P_aff.SkewHoriConfigGeneratorConfig),
    'skew_vert': (P_aff.SkewVertConfigGenerator, P_aff.SkewVertConfigGeneratorConfig),
    'gaussian_blur': (P_blur.GaussianBlurConfigGenerator, P_blur.GaussianBlurConfigGeneratorConfig),
    'mean_shift': (P_color.MeanShiftConfigGenerator, P_color.MeanShiftConfigGeneratorConfig),
    'color_shift': (P_color.ColorShiftConfigGenerator, P_color.ColorShiftConfigGeneratorConfig),
    'gaussion_noise': (P_noise.GaussionNoiseConfigGenerator, P_noise.GaussionNoiseConfigGeneratorConfig),
    'impulse_noise': (P_noise.ImpulseNoiseConfigGenerator, P_noise.ImpulseNoiseConfigGeneratorConfig),
    'speckle_noise': (P_noise.SpeckleNoiseConfigGenerator, P_noise.SpeckleNoiseConfigGeneratorConfig),
    'complement': (P_color.ComplementConfigGenerator, P_color.ComplementConfigGeneratorConfig),
    'fog': (P_effect.FogConfigGenerator, P_effect.FogConfigGeneratorConfig),
    'pixelation': (P_effect.PixelationConfigGenerator, P_effect.PixelationConfigGeneratorConfig),
    'jpeg_quality': (P_effect.JpegQualityConfigGenerator, P_effect.JpegQualityConfigGeneratorConfig),
    'ellipse_streak': (P_streak.EllipseStreakConfigGenerator, P_streak.EllipseStreakConfigGeneratorConfig),
    'glass_blur': (P_blur.GlassBlurConfigGenerator, P_blur.GlassBlurConfigGeneratorConfig),
    'zoom_in_blur': (P_blur.ZoomInBlurConfigGenerator, P_blur.ZoomInBlurConfigGeneratorConfig),
    'boundary_equalization': (P_color.BoundaryEqualizationConfigGenerator,
                              P_color.BoundaryEqualizationConfigGeneratorConfig),
    'histogram_equalization': (P_color.HistogramEqualizationConfigGenerator,
                               P_color.HistogramEqualizationConfigGeneratorConfig),
    'brightness_shift': (P_color.BrightnessShiftConfigGenerator, P_color.BrightnessShiftConfigGeneratorConfig),
    'color_balance': (P_color.ColorBalanceConfigGenerator, P_color.ColorBalanceConfigGeneratorConfig),
    'posterization': (P_color.PosterizationConfigGenerator, P_color.PosterizationConfigGeneratorConfig),
    'channel_permutation': (P_color.ChannelPermutationConfigGenerator,
                            P_color.ChannelPermutationConfigGeneratorConfig),
    'line_streak': (P_streak.LineStreakConfigGenerator, P_streak.LineStreakConfigGeneratorConfig),
    'rectangle_streak': (P_streak.RectangleStreakConfigGenerator, P_streak.RectangleStreakConfigGeneratorConfig),
}


def test_policy_configs_match_reference_draw_for_draw(golden_dir):
    with open(os.path.join(golden_dir, 'policy_configs.json')) as f:
        records = json.load(f)
    checked = 0
    for rec in records:
        assert rec['name'] in GENERATORS, rec['name']
        gen_cls, cfg_cls = GENERATORS[rec['name']]
        rng = default_rng(rec['seed'])
        cfg = gen_cls(cfg_cls(), rec['level'])(tuple(rec['shape']), rng)
        assert plain(cfg) == rec['config'], (rec['name'], rec['level'], rec['seed'])
        # the generator consumed exactly the reference's number of draws
        assert float(rng.random()) == rec['next_random']
        checked += 1
    assert checked == len(records) >= 600


def test_affine_states(golden_dir):
    with open(os.path.join(golden_dir, 'affine_states.json')) as f:
        records = json.load(f)
    classes = {'rotate': (D.geometric.affine.RotateState, D.RotateConfig),
               'shear_hori': (D.geometric.affine.ShearHoriState, D.ShearHoriConfig),
               'shear_vert': (D.geometric.affine.ShearVertState, D.ShearVertConfig)}
    for rec in records:
        if rec['kind'] == 'rotate_points':
            st = D.geometric.affine.RotateState(D.RotateConfig(rec['angle']), (rec['h'], rec['w']), None)
            pts = PointTuple(Point.create(y=y, x=x) for y, x in rec['src'])
            new = D.geometric.affine.affine_points(st.trans_mat, pts)
            assert [[p.smooth_y, p.smooth_x] for p in new] == rec['dst']
            continue
        state_cls, cfg_cls = classes[rec['kind']]
        st = state_cls(cfg_cls(rec['angle']), (rec['h'], rec['w']), None)
        if rec['trans_mat'] is None:
            assert st.trans_mat is None and st.dsize is None and st.result_shape is None
        else:
            assert st.trans_mat.dtype == np.float32
            assert st.trans_mat.astype(np.float64).tolist() == rec['trans_mat'], rec
            assert list(st.dsize) == rec['dsize'], rec
    # SURVEY Appendix B.2
    st = D.geometric.affine.RotateState(D.RotateConfig(30), (512, 512), None)
    assert st.trans_mat.tolist() == [[0.8660253882408142, -0.5, 256.0], [0.5, 0.8660253882408142, 0.0]]
    assert st.dsize == (700, 700) and st.result_shape == (700, 700)


def test_mls_states_bit_identical(golden_dir):
    M = np.load(os.path.join(golden_dir, 'mls_states.npz'))
    meta = json.loads(bytes(M['meta_json']))
    for m in meta:
        k = m['key']
        if k.startswith('big'):
            continue
        cfg = D.SimilarityMlsConfig(
            src_handle_points=PointTuple(Point.create(y=y, x=x) for x, y in M[k + '_src_handles']),
            dst_handle_points=PointTuple(Point.create(y=y, x=x) for x, y in M[k + '_dst_handles']),
            grid_size=m['grid_size'])
        st = D.similarity_mls.generate_state(cfg, (m['h'], m['w']))
        assert st.result_shape == tuple(m['result_shape'])
        assert (st.src_image_grid.vertices == M[k + '_src_grid']).all()
        assert (st.dst_image_grid.smooth == M[k + '_dst_grid_smooth']).all(), k
        assert (st.dst_image_grid.vertices == M[k + '_dst_grid']).all()
        assert [st.shift_amount_y, st.shift_amount_x] == m['shift']


def test_mls_policy_to_state_known_answers(golden_dir):
    """config generator (level 5, seed 0) -> state at 512^2 and 2048^2."""
    M = np.load(os.path.join(golden_dir, 'mls_states.npz'))
    meta = {m['key']: m for m in json.loads(bytes(M['meta_json']))}
    gen = P_mls.SimilarityMlsConfigGenerator(P_mls.SimilarityMlsConfigGeneratorConfig(), 5)
    st = D.similarity_mls.generate_state(gen((512, 512), default_rng(0)), (512, 512))
    assert st.result_shape == (536, 529)
    p = st.dst_image_grid.point(1, 1)
    assert (p.smooth_y, p.smooth_x) == (17.1004319190979, 24.410032272338867)
    big = meta['big2048']
    cfg = gen((2048, 2048), default_rng(0))
    st = D.similarity_mls.generate_state(cfg, (2048, 2048))
    assert cfg.grid_size == 20 and st.src_image_grid.shape == (104, 104) == (big['rows'], big['cols'])
    assert st.result_shape == tuple(big['result_shape']) == (2147, 2115)
    p = st.dst_image_grid.point(1, 1)
    assert [p.smooth_y, p.smooth_x] == big['dst11']
    assert int(st.dst_image_grid.vertices.astype(np.int64).sum()) == big['grid_checksum']


def test_camera_states_structure(golden_dir):
    S = np.load(os.path.join(golden_dir, 'structure_oracle_patched.npz'))
    ops = {'cubic': (D.CameraCubicCurveConfig, D.camera_cubic_curve), 'plane': (D.CameraPlaneOnlyConfig, D.camera_plane_only),
           'fold': (D.CameraPlaneLineFoldConfig, D.camera_plane_line_fold),
           'curve': (D.CameraPlaneLineCurveConfig, D.camera_plane_line_curve)}
    keys = [k[:-len('_dst_grid')] for k in S.files if k.startswith('cam_') and k.endswith('_dst_grid')]
    assert len(keys) == 5
    for key in keys:
        name = key.split('_')[1]
        h, w = (int(v) for v in key.split('_')[2].split('x'))
        cfg_cls, op = ops[name]
        cfg = dyn_structure(json.loads(bytes(S[key + '_config_json'])), cfg_cls)
        st = op.generate_state(cfg, (h, w))
        assert st.result_shape == tuple(S[key + '_result_shape'])
        assert (st.dst_image_grid.smooth == S[key + '_dst_grid_smooth']).all(), key
        assert (st.dst_image_grid.vertices == S[key + '_dst_grid']).all(), key
        pt = S[key + '_point']
        q = FuncImageGridBased.func_point(cfg, st, (h, w), Point.create(y=pt[0], x=pt[1]), None)
        assert (q.smooth_y, q.smooth_x) == (pt[2], pt[3])


def test_camera_theta_zero_keeps_shape():
    # the reference's own assertion (tests/engine/test_camera.py:28-38)
    for vec in ([1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [1.0, 1.0, 0.0]):
        cfg = D.CameraPlaneOnlyConfig(camera_model_config=D.CameraModelConfig(rotation_unit_vec=vec, rotation_theta=0),
                                      grid_size=50)
        st = D.camera_plane_only.generate_state(cfg, (220, 231))
        assert st.result_shape == (220, 231)


def test_random_distortion_table_and_sampling(golden_dir):
    with open(os.path.join(golden_dir, 'random_distortion_sampling.json')) as f:
        ref = json.load(f)
    rd = random_distortion_factory.create(None)
    assert [len(s.config.distortion_policies) for s in rd.stages] == ref['stage_sizes'] == [25, 10]
    assert [[p.name for p in s.config.distortion_policies] for s in rd.stages] == ref['stage_names']
    assert [list(map(float, s.distortion_policy_probs)) for s in rd.stages] == ref['stage_probs']
    assert [s.config.prob_enable for s in rd.stages] == ref['prob_enable'] == [1.0, 0.75]
    for rec in ref['samples']:
        rng = default_rng(rec['seed'])
        names = [[p.name for p in s.sample_distortion_policies(rng)] for s in rd.stages]
        assert names == rec['names']
        assert float(rng.random()) == rec['next_random']
    rd2 = random_distortion_factory.create(RandomDistortionFactoryConfig(
        force_post_rotate=True, disabled_policy_names=['defocus_blur', 'zoom_in_blur']))
    assert [[p.name for p in s.config.distortion_policies] for s in rd2.stages] == ref['post_rotate_stage_names']


def test_glass_shuffle_planes_match_reference(golden_dir):
    """The index planes of glass_blur's pixel shuffle (numpy bookkeeping on the rng stream) against the reference."""
    from vkit_amd.mechanism.distortion.photometric.blur import glass_shuffle_planes
    P = np.load(os.path.join(golden_dir, 'pointwise_ops.npz'))
    for i, (delta, loop, seed) in enumerate(P['glass_cases']):
        pos_y, pos_x = glass_shuffle_planes((97, 141), int(delta), int(loop), default_rng(int(seed)))
        planes = P[f'glass_planes_{i}']
        assert (pos_y == planes[:, :, 0]).all() and (pos_x == planes[:, :, 1]).all(), i
        assert (pos_y != np.arange(97).reshape(-1, 1)).any()


def test_out_of_path_policies_sample_like_the_reference_and_pass_through(golden_dir, monkeypatch, caplog):
    """jpeg_quality: the config is drawn like the reference's (the rng stream stays aligned), the image passes through
    unchanged with one logged warning; VKX_STRICT_UNSUPPORTED=1 raises instead."""
    import logging
    from vkit_amd.mechanism.distortion.photometric import opt as photo_opt
    from vkit_amd.mechanism.distortion.photometric.opt import OUT_OF_PATH_OPERATORS
    assert OUT_OF_PATH_OPERATORS == ('jpeg_quality',)
    with open(os.path.join(golden_dir, 'policy_configs.json')) as f:
        records = [r for r in json.load(f) if r['name'] in OUT_OF_PATH_OPERATORS]
    assert {r['name'] for r in records} == set(OUT_OF_PATH_OPERATORS)
    rd = random_distortion_factory.create(None)
    image = Image(mat=default_rng(0).integers(0, 256, (96, 80, 3), dtype=np.uint8))
    for name in OUT_OF_PATH_OPERATORS:
        policy = [p for p in rd.stages[0].config.distortion_policies if p.name == name][0]
        rec = [r for r in records if r['name'] == name and r['level'] == 5 and r['seed'] == 1 and r['shape'] == [96, 80]][0]
        rng = default_rng(1)
        photo_opt._warned.discard(name)
        with caplog.at_level(logging.WARNING):
            res = policy.distort(level=5, image=image, rng=rng, enable_debug=True)
        assert plain(res.config) == rec['config']
        assert float(rng.random()) == rec['next_random']          # exactly the reference's draws
        assert (res.image.mat == image.mat).all()
        assert any(name in r.getMessage() for r in caplog.records)
        assert res.meta == {'out_of_path': (name,)}              # ... and the result says that a stage was not applied
        monkeypatch.setenv('VKX_STRICT_UNSUPPORTED', '1')
        with pytest.raises(NotImplementedError):
            policy.distort(level=5, image=image, rng=default_rng(1))
        monkeypatch.delenv('VKX_STRICT_UNSUPPORTED')
        with photo_opt.out_of_path('raise'), pytest.raises(NotImplementedError):
            policy.distort(level=5, image=image, rng=default_rng(1))
    # the factory switch: a RandomDistortion that refuses to skip a stage, and one that records what it skipped (only
    # jpeg_quality left in the table: no GPU in this test)
    others = [f.name for f in random_distortion_factory.photometric_policy_factories if f.name != 'jpeg_quality']
    config = {'prob_photometric': 1.0, 'num_photometric_min': 1, 'num_photometric_max': 1, 'prob_geometric': 0.0,
              'disabled_policy_names': others}
    lenient = random_distortion_factory.create(config)
    out = lenient.distort(default_rng(3), image=image)
    assert (out.image.mat == image.mat).all() and out.meta == {'out_of_path': ('jpeg_quality',)}
    strict = random_distortion_factory.create(config, out_of_path='raise')
    with pytest.raises(NotImplementedError):
        strict.distort(default_rng(3), image=image)
    with photo_opt.out_of_path('raise'), pytest.raises(NotImplementedError):
        lenient.distort(default_rng(3), image=image)
    with photo_opt.out_of_path('raise'):                 # the factory's own choice wins over the surrounding one
        assert random_distortion_factory.create(config, out_of_path='pass_through').distort(default_rng(3), image=image).meta


def test_default_random_distortion_never_raises_on_sampling():
    """the default policy table must be safe -- every policy of the default factory config resolves to a
    runnable operator (no raising placeholder), over many seeds of the sampler."""
    rd = random_distortion_factory.create()
    seen = set()
    for seed in range(400):
        rng = default_rng(seed)
        for stage in rd.stages:
            for policy in stage.sample_distortion_policies(rng):
                seen.add(policy.name)
                assert callable(getattr(policy, 'distort', None)) and policy.distortion.func_image is not None
    assert {'jpeg_quality', 'ellipse_streak'} <= seen


def test_operator_geometry_without_pixels(golden_dir):
    with open(os.path.join(golden_dir, 'operator_semantics.json')) as f:
        ref = json.load(f)['rotate_op']
    pts = PointList([Point.create(y=1.5, x=2.5), Point.create(y=15, x=15), Point.create(y=0, x=15)])
    poly = Polygon.create(points=[Point.create(y=2, x=2), Point.create(y=2, x=12), Point.create(y=12, x=12)])
    res = D.rotate.distort({'angle': 33}, shapable_or_shape=(16, 16), points=pts, polygon=poly, corner_points=pts,
                           get_state=True)
    assert list(res.shape) == ref['shape']
    assert [[p.smooth_y, p.smooth_x] for p in res.points] == ref['points']
    assert [[p.smooth_y, p.smooth_x] for p in res.corner_points] == ref['corner_points']
    assert [[p.smooth_y, p.smooth_x] for p in res.polygon.points] == ref['polygon']
    assert res.state is not None and res.config is None


def test_operator_rng_contract():
    """Reference interface.py:261-307: capture state, advance the caller once, rewind before every element."""

    @attrs.define
    class ProbeConfig(DistortionConfig):
        scale: float = 1.0
        _rng_state: object = None

        @property
        def supports_rng_state(self):
            return True

        @property
        def rng_state(self):
            return self._rng_state

        @rng_state.setter
        def rng_state(self, val):
            self._rng_state = val

    draws = []

    def func_image(config, state, image, rng):
        draws.append(('image', rng.random()))
        return image

    def func_mask(config, state, mask, rng):
        draws.append(('mask', rng.random()))
        return mask

    probe = Distortion(ProbeConfig, DistortionNopState[ProbeConfig], func_image, func_mask=func_mask)
    img = Image(mat=np.zeros((4, 4, 3), np.uint8))
    mask = Mask(mat=np.zeros((4, 4), np.uint8))
    score = ScoreMap(mat=np.zeros((4, 4), np.float32))

    rng = default_rng(7)
    expect_first = default_rng(7).random()
    res = probe.distort(lambda shape, r: ProbeConfig(scale=r.uniform()), image=img, mask=mask, score_map=score, rng=rng,
                        get_config=True)
    # generator's uniform came first, so the captured state is the one after it
    g = default_rng(7)
    scale = g.uniform()
    assert res.config.scale == scale and scale == expect_first
    first_private = g.random()
    assert draws == [('image', first_private), ('mask', first_private)]  # same stream for every element
    assert float(rng.random()) == float(g.random())  # caller advanced by exactly one extra random()
    assert res.score_map is score and res.state is None and res.shape == (4, 4)  # missing func -> same object
    # replay from the stored config needs no rng
    draws.clear()
    probe.distort_image(res.config, img)
    assert draws == [('image', first_private)]
    with pytest.raises(RuntimeError):
        probe.distort_image(ProbeConfig(), img)
    with pytest.raises(RuntimeError):
        probe.distort_image(lambda shape, r: ProbeConfig(), img)


def test_dyn_structure_rejects_unknown_keys_and_nests():
    cfg = dyn_structure({'camera_model_config': {'rotation_unit_vec': [1, 0, 0], 'rotation_theta': 3}, 'grid_size': 20},
                        D.CameraPlaneOnlyConfig)
    assert isinstance(cfg.camera_model_config, D.CameraModelConfig) and cfg.grid_size == 20
    with pytest.raises(TypeError):
        dyn_structure({'angle': 3, 'bogus': 1}, D.RotateConfig)
    assert dyn_structure(D.RotateConfig(3), D.RotateConfig).angle == 3
    ms = dyn_structure({'delta': 5, 'oob_behavior': 'cycle'}, D.MeanShiftConfig)
    assert ms.oob_behavior is D.OutOfBoundBehavior.CYCLE


def test_element_basics():
    assert D.RotateConfig.get_name() == 'rotate' and D.CameraCubicCurveConfig.get_name() == 'camera_cubic_curve'
    assert D.rotate.is_geometric and D.similarity_mls.is_geometric and not D.gaussian_blur.is_geometric
    p = Point.create(y=2.5, x=3.5)
    assert (p.y, p.x) == (2, 4)  # half to even
    assert Point.create(y=1.2, x=7) == Point.create(y=0.9, x=7.4)
    t = PointTuple([Point.create(y=1.4, x=2.6)])
    assert t.to_smooth_np_array().tolist() == [[3.0, 1.0]]  # integer positions: the reference's quirk
    assert PointList(t).to_smooth_np_array().tolist() == [[np.float32(2.6), np.float32(1.4)]]
    img = Image(mat=np.zeros((5, 7, 3), np.uint8))
    assert img.mode is ImageMode.RGB and img.shape == (5, 7) and not img.mat.flags.writeable
    with img.writable_context:
        img.mat[0, 0, 0] = 9
    assert img.mat[0, 0, 0] == 9 and not img.mat.flags.writeable
    with pytest.raises(RuntimeError):
        ScoreMap(mat=np.full((2, 2), 2.0, np.float32))
    with pytest.raises(RuntimeError):
        Mask(mat=np.zeros((2, 2), np.float32))
    box = Box(up=1, down=3, left=2, right=5)
    assert box.shape == (3, 4) and box.to_polygon().bounding_box == box
    assert Mask.from_shape((3, 3), value=1).to_inverted_mask().mat.sum() == 0
    from vkit_amd.mechanism.distortion.photometric.blur import _estimate_gaussian_kernel_size as ks
    assert [ks(s) for s in (0.5, 0.83, 0.84, 1.0, 2.0)] == [3, 3, 5, 5, 7]


def test_page_resizing_step_decisions(golden_dir):
    """PageResizingStep against the reference's own run on recording stand-ins (tests/golden/make_golden.py
    gen_page_resizing): text-line height floor, rng order, target size, interpolation, element order, score scaling."""
    from types import SimpleNamespace
    from vkit_amd.pipeline.text_detection import PageResizingStep, PageResizingStepConfig, PageResizingStepInput
    from vkit_amd.utility import sample_cv_resize_interpolation
    with open(os.path.join(golden_dir, 'page_resizing.json')) as f:
        G = json.load(f)
    for s, (plain, with_area) in enumerate(G['interpolation_draws']):
        assert sample_cv_resize_interpolation(default_rng(s), False) == plain
        assert sample_cv_resize_interpolation(default_rng(s), True) == with_area

    class Recorder:
        def __init__(self, name, shape, log):
            self.name, self.shape, self.log, self.mat = name, shape, log, 1.0

        def _resized(self, **kwargs):
            self.log.append([self.name, kwargs['resized_height'], kwargs['resized_width'], kwargs['cv_resize_interpolation']])
            return Recorder(self.name, (kwargs['resized_height'], kwargs['resized_width']), self.log)

        to_resized_image = to_resized_mask = to_resized_score_map = _resized

        def assign_mat(self, mat):
            self.log.append([self.name + '.scale', float(mat)])

    assert len(G['cases']) == 40
    seen = set()
    for case in G['cases']:
        lo, hi, thr = case['config']
        step = PageResizingStep(PageResizingStepConfig(resized_text_line_height_min=lo, resized_text_line_height_max=hi,
                                                       text_line_heights_filtering_thr=thr))
        assert step.get_text_line_heights_min(case['heights']) == case['heights_min']
        log = []
        page = SimpleNamespace(page_text_line_heights=case['heights'])
        for name in ('page_image', 'page_active_mask', 'page_char_mask', 'page_seal_impression_char_mask',
                     'page_char_height_score_map', 'page_text_line_mask', 'page_text_line_height_score_map'):
            setattr(page, name, Recorder(name, tuple(case['shape']), log))
        out = step.run(PageResizingStepInput(page_distortion_step_output=page), default_rng(case['seed']))
        assert log == case['calls'], case['seed']
        assert out.page_image.shape == (case['calls'][0][1], case['calls'][0][2])
        seen.add(case['calls'][0][3])
    assert seen == {2, 3, 4, 5, 6}      # every interpolation the step can draw shows up in the fixture


def _std_shift_cases(golden_dir):
    Z = np.load(os.path.join(golden_dir, 'std_shift.npz'))
    cases = json.loads(bytes(Z['cases_json']))
    for i, case in enumerate(cases):
        shape = tuple(case['shape'])
        if f'in_{i}' in Z.files:
            yield case, Z[f'in_{i}'], Z[f'out_{i}'], None, None
        else:
            n = int(np.prod(shape))
            mat = (np.arange(n, dtype=np.uint32) * np.uint32(2654435761) >> np.uint32(24)).astype(np.uint8).reshape(shape)
            yield case, mat, None, Z[f'out_hist_{i}'], Z[f'out_head_{i}']


def test_std_shift_tables_match_reference(golden_dir, monkeypatch):
    """std_shift's host half (numpy mean + the 256-level float32 expression) against the reference's outputs; the table
    pass itself is replaced by numpy indexing here and runs on the GPU in tests/test_gpu_pointwise.py."""
    from vkit_amd import _native

    def lut_on_host(img, lut, channels=None, ctx=None):
        out = img.copy()
        planes = out.reshape(out.shape[0], out.shape[1], -1)
        for c in (channels if channels is not None else range(planes.shape[2])):
            planes[:, :, c] = lut[c][planes[:, :, c]]
        return out

    monkeypatch.setattr(_native, 'apply_lut', lut_on_host)
    monkeypatch.setenv('VKX_HOST_MEAN', '1')     # no GPU here: numpy's mean (the device's is checked against numpy in test_gpu_reduce.py)
    n = 0
    for case, mat, want, hist, head in _std_shift_cases(golden_dir):
        got = D.std_shift.distort(D.StdShiftConfig(scale=case['scale'], channels=case['channels']), image=Image(mat=mat)).image.mat
        if want is not None:
            assert (got == want).all(), case
        else:
            assert (np.bincount(got.reshape(-1), minlength=256) == hist).all(), case
            assert (got.reshape(-1)[:4096] == head).all(), case
        n += 1
    assert n == 9


def test_layer_records_of_the_composite(monkeypatch):
    """What ``Box.fill_*`` hands to the composite (no GPU: the records are collected by an open deferred composite and the
    block is abandoned): geometry, plane addresses, strides and modes; a plain uint8 alpha blend carries no selection plane
    ((1 - 0) * d + 0 * v is d exactly), keep_max / float32 destinations keep the reference's ``alpha > 0`` selection."""
    from vkit_amd import _native as N
    from vkit_amd.element import Box, Image, Mask, ScoreMap
    from vkit_amd.element.opt import deferred_fill
    rng = default_rng(5)
    page = Image(mat=rng.integers(0, 256, (40, 60, 3), dtype=np.uint8))
    alpha = (rng.random((8, 10), dtype=np.float32) * (rng.random((8, 10)) < 0.5)).astype(np.float32)
    box = Box(up=3, down=10, left=7, right=16)
    value = rng.integers(0, 256, (8, 10, 3), dtype=np.uint8)
    mask = Mask(mat=(rng.random((8, 10)) < 0.5).astype(np.uint8), box=box)

    class Abandon(Exception):
        pass

    with pytest.raises(Abandon):
        with page.writable_context, deferred_fill(page.mat) as session:
            ScoreMap(mat=alpha, box=box).fill_image(page, (1, 2, 3))
            box.fill_image(page, Image(mat=value), alpha=0.25)
            mask.fill_image(page, Image(mat=value))
            box.fill_np_array(page.mat, 200, alpha=alpha, keep_max_value=True)
            recorded = list(session.layers)
            raise Abandon()
    assert len(recorded) == 4
    (l0, k0), (l1, k1), (l2, k2), (l3, k3) = recorded
    for layer in (l0, l1, l2, l3):
        assert (layer.up, layer.left, layer.height, layer.width) == (3, 7, 8, 10)
    # score map, constant colour: alpha plane, no mask, the colour in value_const
    assert l0.alpha == alpha.__array_interface__['data'][0] and l0.alpha_stride_el == 10 and not l0.mask and not l0.value
    assert list(l0.value_const)[:3] == [1, 2, 3] and l0.mode == N.FILL_PLAIN and any(p is alpha for p in k0)
    # image value, scalar alpha
    assert l1.value == value.__array_interface__['data'][0] and l1.value_stride == 30 and l1.alpha_scalar == 0.25 and not l1.alpha
    # mask selects, value plane
    assert l2.mask and l2.mask_stride == 10 and l2.value and l2.alpha_scalar == 1.0
    # keep_max with an alpha plane keeps the reference's alpha > 0 selection
    assert l3.mode == N.FILL_KEEP_MAX and l3.mask and l3.alpha
    assert (k3[0] == (alpha > 0)).all()
    # the page was not written and is read-only again
    assert not page.mat.flags.writeable
    # a float32 destination keeps the selection plane too
    smap = ScoreMap(mat=np.zeros((40, 60), np.float32))
    with pytest.raises(Abandon):
        with smap.writable_context, deferred_fill(smap.mat) as session:
            box.fill_np_array(smap.mat, 0.5, alpha=alpha)
            assert session.layers[0][0].mask
            raise Abandon()
    # constants outside uint8 fail like numpy's own assignment does
    with pytest.raises(OverflowError):
        N.make_layer((0, 0, 2, 2), 3, (300, 0, 0))
    layer, keep = N.make_layer((0, 0, 2, 2), 1, 7.9)
    assert layer.value_const[0] == 7


def test_direct_layer_records_equal_the_generic_path():
    """``ScoreMap.fill_image`` / ``Box.fill_image`` / ``Mask.fill_image`` build the record of an open deferred composite
    directly; the generic path (``Box._fill_element`` -> ``Box.fill_np_array`` -> ``opt.fill_np_array``) must produce the very
    same bytes, and whatever the direct path does not take (an image with a box, a wrong plane, an int alpha) reaches it."""
    import ctypes
    from vkit_amd.element import Box, Image, Mask, ScoreMap
    from vkit_amd.element.opt import deferred_fill
    rng = default_rng(8)
    page = Image(mat=rng.integers(0, 256, (50, 70, 3), dtype=np.uint8))
    box = Box(up=5, down=16, left=9, right=28)
    alpha = rng.random(box.shape, dtype=np.float32)
    value = Image(mat=rng.integers(0, 256, box.shape + (3,), dtype=np.uint8))
    mask = Mask(mat=(rng.random(box.shape) < 0.5).astype(np.uint8), box=box)
    smap = ScoreMap(mat=alpha, box=box)

    class Abandon(Exception):
        pass

    def record(fn):
        with pytest.raises(Abandon):
            with page.writable_context, deferred_fill(page.mat) as session:
                fn()
                layers = []
                for layer, keep in session.layers:
                    # the selection plane: the generic path hands over a fresh boolean copy, the direct one the mask's own
                    # bytes -- the composite selects where the byte is non-zero, so compare that
                    selection = None
                    if layer.mask:
                        plane = next(p for p in keep if p.__array_interface__['data'][0] == layer.mask)
                        selection = (np.asarray(plane) > 0).tobytes()
                        layer.mask = 1
                    layers.append((bytes(ctypes.string_at(ctypes.addressof(layer), ctypes.sizeof(layer))), selection))
                raise Abandon()
        return layers

    direct = record(lambda: (smap.fill_image(page, (1, 2, 3)), box.fill_image(page, value, alpha=0.5),
                             box.fill_image(page, (9, 8, 7), image_mask=mask, alpha=1.0), mask.fill_image(page, value),
                             box.fill_image(page, value, alpha=smap)))
    generic = record(lambda: (box._fill_element(page, (1, 2, 3), Image, None, alpha=smap),
                              box._fill_element(page, value, Image, None, alpha=0.5),
                              box._fill_element(page, (9, 8, 7), Image, mask, alpha=1.0),
                              box._fill_element(page, value, Image, mask, alpha=1.0),
                              box._fill_element(page, value, Image, None, alpha=smap)))
    assert len(direct) == 5 and direct == generic
    # not taken by the direct path: errors are the generic path's
    with pytest.raises(AttributeError):
        record(lambda: box.fill_image(page, value, alpha=1))
    with pytest.raises(RuntimeError):
        record(lambda: box.fill_image(page, value, alpha=1.5))
    attached = Image(mat=page.mat, box=Box(up=100, down=149, left=200, right=269))
    with pytest.raises(Abandon):
        with attached.writable_context, deferred_fill(attached.mat) as session:
            ScoreMap(mat=alpha, box=Box(up=105, down=116, left=209, right=228)).fill_image(attached, (1, 2, 3))
            layer = session.layers[0][0]
            assert (layer.up, layer.left, layer.height, layer.width) == (5, 9, 12, 20)
            raise Abandon()
