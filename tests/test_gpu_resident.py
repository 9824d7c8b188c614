"""Device-resident arrays: every wrapper that takes a DevArray gives the bytes of its host-array path, elements hold DevArrays
behind a lazy ``.mat``, and a chain of operators in resident mode equals the same chain on host arrays."""
import numpy as np
import pytest

from vkit_amd import _native as N
from vkit_amd.element import Image, Mask, ScoreMap
from vkit_amd.mechanism import distortion as D
from vkit_amd.mechanism.distortion_policy import random_distortion_factory

pytestmark = pytest.mark.gpu


def _rgb(seed, h=203, w=257):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def test_wrappers_on_device_arrays_match_their_host_path():
    ctx = N.default_ctx()
    img = _rgb(0)
    gray = img[:, :, 0].copy()
    score = np.random.default_rng(1).random(img.shape[:2], dtype=np.float32)
    dev, dgray, dscore = ctx.to_device(img), ctx.to_device(gray), ctx.to_device(score)
    lut = np.random.default_rng(2).integers(0, 256, (3, 256), dtype=np.uint8)
    noise = np.random.default_rng(3).integers(-40, 40, img.shape).astype(np.int16)
    kernel = np.random.default_rng(4).random((5, 7)).astype(np.float32)
    M = np.array([[0.9, -0.2, 10.0], [0.15, 1.1, -4.0]], np.float32)
    cases = [
        lambda a: N.gaussian_blur(a, 5, 1.0), lambda a: N.color_shift_rgb(a, 37), lambda a: N.cvt_rgb_hsv(a, True),
        lambda a: N.mean_shift(a, -30, threshold=100, channels=[1]), lambda a: N.cvt_color(a, N.CVT_RGB2GRAY),
        lambda a: N.cvt_color(a, N.CVT_RGB2HLS_FULL), lambda a: N.brightness_shift_rgb(a, 25),
        lambda a: N.color_balance_rgb(a, 0.4), lambda a: N.pointwise(a, N.POINT_COMPLEMENT, -1, 0),
        lambda a: N.apply_lut(a, lut), lambda a: N.filter2d(a, kernel), lambda a: N.add_noise_i16(a, noise),
        lambda a: N.resize(a, (150, 311), N.INTER_CUBIC), lambda a: N.resize(a, (411, 200), N.INTER_LANCZOS4),
        lambda a: N.resize_cubic(a, (99, 120)), lambda a: N.warp_affine(a, M, (300, 240)),
        lambda a: N.line_streak(a, 2, 9, 3, 2, (10, 200, 30), 0.4, True, True),
        lambda a: N.ellipse_streak(a, (128, 100), [(40, 30), (80, 60)], 2, (1, 2, 3), 0.7),
        lambda a: N.blend_u8(a, img[::-1].copy(), 0.3, 0.7), lambda a: N.zoom_in_blur(a, [(220, 280), (260, 330)], 0.5),
    ]
    for k, fn in enumerate(cases):
        want, got = fn(img), fn(dev)
        assert isinstance(got, N.DevArray), k
        assert got.shape == want.shape and (got.host() == want).all(), k
    assert (N.histogram(dev) == N.histogram(img)).all()
    assert (N.resize(dscore, (100, 90), N.INTER_LINEAR).host() == N.resize(score, (100, 90), N.INTER_LINEAR)).all()
    assert (N.cvt_color(dgray, N.CVT_GRAY2RGB).host() == N.cvt_color(gray, N.CVT_GRAY2RGB)).all()
    # resident mode: host arrays in, DevArrays out
    with N.resident():
        out = N.gaussian_blur(img, 3, 0.7)
    assert isinstance(out, N.DevArray) and (out.host() == N.gaussian_blur(img, 3, 0.7)).all()


def test_elements_hold_device_arrays_lazily():
    ctx = N.default_ctx()
    img = _rgb(5)
    image = Image(mat=ctx.to_device(img))
    assert image.on_device and image.shape == img.shape[:2] and image.mode.name == 'RGB'
    blurred = D.gaussian_blur.distort({'sigma': 1.0}, image=image).image
    assert blurred.on_device                                    # device in, device out: nothing crossed the link
    assert (blurred.mat == D.gaussian_blur.distort({'sigma': 1.0}, image=Image(mat=img)).image.mat).all()
    assert not blurred.mat.flags.writeable
    with blurred.writable_context:                              # writing moves the element to the host
        blurred.mat[0, 0] = 7
    assert not blurred.on_device and blurred.mat[0, 0, 0] == 7
    mask = Mask(mat=ctx.to_device((img[:, :, 0] > 100).astype(np.uint8)))
    assert (mask.to_inverted_mask().mat == (img[:, :, 0] <= 100)).all()
    host_mask = Mask(mat=(img[:, :, 0] > 100).astype(np.uint8))
    assert (mask.to_resized_mask(resized_height=77, resized_width=91).mat
            == host_mask.to_resized_mask(resized_height=77, resized_width=91).mat).all()
    score = ScoreMap(mat=ctx.to_device(np.random.default_rng(6).random(img.shape[:2], dtype=np.float32)), is_prob=False)
    assert score.to_resized_score_map(resized_height=50).on_device


@pytest.mark.parametrize('seed', range(12))
def test_random_distortion_resident_equals_host(seed):
    """The default policy table on a page, operators chained on the device vs. every operator through host arrays."""
    img = _rgb(100 + seed, 256, 320)
    mask = np.ones(img.shape[:2], np.uint8)
    mask[:3] = 0
    score = np.random.default_rng(seed).random(img.shape[:2], dtype=np.float32)
    rd = random_distortion_factory.create()        # poisson_noise included: drawn on the device since round 4
    host = rd.distort(np.random.default_rng(seed), image=Image(mat=img), mask=Mask(mat=mask),
                      score_map=ScoreMap(mat=score, is_prob=False))
    with N.resident():
        dev = rd.distort(np.random.default_rng(seed), image=Image(mat=img), mask=Mask(mat=mask),
                         score_map=ScoreMap(mat=score, is_prob=False))
    assert dev.shape == host.shape
    assert (dev.image.mat == host.image.mat).all()
    assert (dev.mask.mat == host.mask.mat).all()
    assert (dev.score_map.mat.view(np.uint32) == host.score_map.mat.view(np.uint32)).all()


def _policy_names():
    f = random_distortion_factory
    return [p.name for p in f.photometric_policy_factories + f.geometric_policy_factories]


@pytest.mark.parametrize('name', _policy_names())
def test_every_policy_of_the_table_in_resident_mode(name):
    """Each policy of the default table once, explicitly (a random page draws the rare ones too seldom to trust): in resident
    mode -- the way ``PageDistortionStep.run`` drives them -- the result equals the host-array run, elements and polygons
    included, and helpers that post-process a wrapper result with numpy (the motion-blur kernel, ``Polygon.np_mask`` behind the
    active mask) keep working."""
    from vkit_amd.element import Polygon
    f = random_distortion_factory
    factory = next(p for p in f.photometric_policy_factories + f.geometric_policy_factories if p.name == name)
    policy = factory.create(None)
    img = _rgb(7, 192, 224)
    mask = (np.random.default_rng(8).random(img.shape[:2]) < 0.5).astype(np.uint8)
    score = np.random.default_rng(9).random(img.shape[:2], dtype=np.float32)
    polygon = Polygon.from_xy_pairs([(20, 30), (150, 25), (170, 120), (40, 140)])
    outs = []
    for res in (False, True):
        with N.resident(res):
            for level in (3, 9):
                r = policy.distort(level, (img.shape[0], img.shape[1]), rng=np.random.default_rng(50 + level), image=Image(mat=img),
                                   mask=Mask(mat=mask), score_map=ScoreMap(mat=score, is_prob=False), polygon=polygon,
                                   get_active_mask=True)
                outs.append((res, level, r))
    for (_, level, host), (_, _, dev) in zip(outs[:2], outs[2:]):
        assert dev.shape == host.shape
        assert (dev.image.mat == host.image.mat).all(), (name, level)
        assert (dev.mask.mat == host.mask.mat).all(), (name, level)
        assert (dev.score_map.mat.view(np.uint32) == host.score_map.mat.view(np.uint32)).all(), (name, level)
        assert (dev.active_mask.mat == host.active_mask.mat).all(), (name, level)
        assert (dev.polygon.to_np_array() == host.polygon.to_np_array()).all(), (name, level)
