"""complement / posterization / channel_permutation / impulse_noise / speckle_noise through the operator API on the
GPU, against genuine reference outputs (tests/golden/pointwise_ops.npz) and against the oracle at full size."""
import json
import os

import numpy as np
import pytest
from numpy.random import default_rng

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def P(golden_dir):
    return np.load(os.path.join(golden_dir, 'pointwise_ops.npz'))


def test_operators_reproduce_reference_outputs(P):
    from vkit_amd.element import Image
    from vkit_amd.mechanism import distortion as D
    src = P['src']
    img = Image(mat=src)
    for i, kw in enumerate(json.loads(str(P['complement_cases']))):
        np.testing.assert_array_equal(D.complement.distort(kw, image=img).image.mat, P[f'complement_{i}'])
    for bits in range(8):
        np.testing.assert_array_equal(D.posterization.distort({'num_bits': bits}, image=img).image.mat,
                                      P[f'posterization_{bits}'])
    np.testing.assert_array_equal(D.posterization.distort({'num_bits': 3, 'channels': [1]}, image=img).image.mat,
                                  P['posterization_3_c1'])
    for seed in (0, 1, 2, 3):
        np.testing.assert_array_equal(D.channel_permutation.distort({}, image=img, rng=default_rng(seed)).image.mat,
                                      P[f'channel_permutation_{seed}'])
    for i, (ps, pp, seed) in enumerate(P['impulse_cases']):
        out = D.impulse_noise.distort({'prob_salt': float(ps), 'prob_pepper': float(pp)}, image=img,
                                      rng=default_rng(int(seed))).image.mat
        np.testing.assert_array_equal(out, P[f'impulse_{i}'])
    for i, (std, seed) in enumerate(P['speckle_cases']):
        out = D.speckle_noise.distort({'std': float(std)}, image=img, rng=default_rng(int(seed))).image.mat
        np.testing.assert_array_equal(out, P[f'speckle_{i}'])
    gray = Image(mat=src[:, :, 0].copy())
    np.testing.assert_array_equal(D.complement.distort({'threshold': 128}, image=gray).image.mat,
                                  P['gray_complement_thr'])
    np.testing.assert_array_equal(
        D.impulse_noise.distort({'prob_salt': 0.1, 'prob_pepper': 0.1}, image=gray, rng=default_rng(5)).image.mat,
        P['gray_impulse'])
    np.testing.assert_array_equal(D.speckle_noise.distort({'std': 0.2}, image=gray, rng=default_rng(6)).image.mat,
                                  P['gray_speckle'])


def test_rng_state_replay(P):
    """A config that went through distort() carries the generator state: replaying it reproduces the pixels."""
    from vkit_amd.element import Image
    from vkit_amd.mechanism import distortion as D
    img = Image(mat=P['src'])
    for op, kw in ((D.impulse_noise, {'prob_salt': 0.05, 'prob_pepper': 0.05}), (D.speckle_noise, {'std': 0.2}),
                   (D.channel_permutation, {})):
        first = op.distort(kw, image=img, rng=default_rng(42), get_config=True)
        again = op.distort(first.config, image=img)
        np.testing.assert_array_equal(first.image.mat, again.image.mat)


def test_full_size_against_oracle():
    from vkit_amd import _native as N
    rng = default_rng(12)
    src = rng.integers(0, 256, (2048, 2048, 3), dtype=np.uint8)
    np.testing.assert_array_equal(N.pointwise(src, N.POINT_COMPLEMENT, 90, 1, channels=[0, 2]),
                                  O.complement(src, 90, True, [0, 2]))
    np.testing.assert_array_equal(N.pointwise(src, N.POINT_POSTERIZE, 5), O.posterization(src, 5))
    np.testing.assert_array_equal(N.permute_channels(src, [2, 0, 1]), src[:, :, [2, 0, 1]])
    sel = rng.choice((0, 1, 2), size=src.shape[:2], p=[0.9, 0.04, 0.06]).astype(np.uint8)
    np.testing.assert_array_equal(N.impulse_noise(src, sel), O.impulse_noise(src, sel))
    noise = rng.normal(0, 0.3, src.shape)
    np.testing.assert_array_equal(N.speckle_noise(src, noise), O.speckle_noise(src, noise))
    four = rng.integers(0, 256, (33, 47, 4), dtype=np.uint8)
    np.testing.assert_array_equal(N.permute_channels(four, [3, 1, 0, 2]), four[:, :, [3, 1, 0, 2]])


def test_bad_arguments_are_refused():
    from vkit_amd import _native as N
    src = np.zeros((4, 4, 3), np.uint8)
    with pytest.raises(N.VkxError):
        N.pointwise(src, N.POINT_POSTERIZE, 9)
    with pytest.raises(N.VkxError):
        N.pointwise(src, 7)
    with pytest.raises(N.VkxError):
        N.pointwise(src, N.POINT_PERMUTE, 0b111111)  # index 3 on a 3-channel image


def _rgb_cube():
    v = np.arange(256, dtype=np.uint8)
    return np.stack(np.meshgrid(v, v, v, indexing='ij'), axis=-1).reshape(4096, 4096, 3)


def test_hls_gray_conversions_exhaustive():
    """Every 24-bit colour through RGB2HLS_FULL, HLS2RGB_FULL, RGB2GRAY, brightness_shift and color_balance."""
    from vkit_amd import _native as N
    cube = _rgb_cube()
    np.testing.assert_array_equal(N.cvt_color(cube, N.CVT_RGB2HLS_FULL), O.rgb2hls_full(cube))
    np.testing.assert_array_equal(N.cvt_color(cube, N.CVT_HLS2RGB_FULL), O.hls2rgb_full(cube))
    np.testing.assert_array_equal(N.cvt_color(cube, N.CVT_RGB2GRAY), O.rgb2gray(cube))
    for delta in (0, 1, -37, 127, -127):
        np.testing.assert_array_equal(N.brightness_shift_rgb(cube, delta), O.brightness_shift_rgb(cube, delta))
    for ratio in (0.0, 0.123456789, 0.5, 0.999, 1.0):
        np.testing.assert_array_equal(N.color_balance_rgb(cube, ratio), O.color_balance_rgb(cube, ratio))
    gray = default_rng(1).integers(0, 256, (37, 41), dtype=np.uint8)
    np.testing.assert_array_equal(N.cvt_color(gray, N.CVT_GRAY2RGB), np.repeat(gray[:, :, None], 3, axis=2))
    with pytest.raises(N.VkxError):
        N.color_balance_rgb(cube[:4, :4], 1.5)


def test_image_modes_and_operators():
    from vkit_amd.element import Image, ImageMode
    from vkit_amd.mechanism import distortion as D
    rng = default_rng(3)
    rgb = Image(mat=rng.integers(0, 256, (45, 67, 3), dtype=np.uint8))
    hsl = rgb.to_hsl_image()
    assert hsl.mode == ImageMode.HSL
    np.testing.assert_array_equal(hsl.mat, O.rgb2hls_full(rgb.mat)[:, :, [0, 2, 1]])
    np.testing.assert_array_equal(hsl.to_rgb_image().mat, O.hls2rgb_full(O.rgb2hls_full(rgb.mat)))
    gray = rgb.to_grayscale_image()
    assert gray.mode == ImageMode.GRAYSCALE and gray.mat.ndim == 2
    np.testing.assert_array_equal(gray.mat, O.rgb2gray(rgb.mat))
    np.testing.assert_array_equal(gray.to_hsv_image().mat,
                                  O.rgb2hsv_full(np.repeat(O.rgb2gray(rgb.mat)[:, :, None], 3, axis=2)))
    # brightness_shift: fused RGB path == the reference's three-step path through an HSL image
    out = D.brightness_shift.distort({'delta': 40}, image=rgb).image
    np.testing.assert_array_equal(out.mat, O.brightness_shift_rgb(rgb.mat, 40))
    stepwise = O.mean_shift(hsl.mat, 40, channels=[2])
    np.testing.assert_array_equal(D.brightness_shift.distort({'delta': 40}, image=hsl).image.mat, stepwise)
    np.testing.assert_array_equal(out.mat, O.hls2rgb_full(stepwise[:, :, [0, 2, 1]]))
    # HSV intermediate on request
    hsv_way = D.brightness_shift.distort({'delta': -25, 'intermediate_image_mode': 'hsv'}, image=rgb).image
    np.testing.assert_array_equal(hsv_way.mat,
                                  O.hsv2rgb_full(O.mean_shift(O.rgb2hsv_full(rgb.mat), -25, channels=[2])))
    out = D.color_balance.distort({'ratio': 0.3}, image=rgb).image
    np.testing.assert_array_equal(out.mat, O.color_balance_rgb(rgb.mat, 0.3))
    assert D.color_balance.distort({'ratio': 0.3}, image=gray).image is gray


def test_equalisations(P):
    from vkit_amd import _native as N
    from vkit_amd.element import Image
    from vkit_amd.mechanism import distortion as D
    low = P['beq_src']
    # histogram / table primitives
    hist = N.histogram(low)
    for c in range(3):
        np.testing.assert_array_equal(hist[c], np.bincount(low[:, :, c].ravel(), minlength=256))
    lut = np.stack([np.roll(np.arange(256, dtype=np.uint8), 7 * (c + 1)) for c in range(3)])
    want = low.copy()
    for c in (0, 2):
        want[:, :, c] = lut[c][low[:, :, c]]
    np.testing.assert_array_equal(N.apply_lut(low, lut, channels=[0, 2]), want)
    big = default_rng(8).integers(0, 256, (2048, 2048, 3), dtype=np.uint8)
    hist = N.histogram(big)
    for c in range(3):
        np.testing.assert_array_equal(hist[c], np.bincount(big[:, :, c].ravel(), minlength=256))
    # boundary_equalization: genuine reference outputs (numpy only: pinned)
    limg = Image(mat=low)
    np.testing.assert_array_equal(D.boundary_equalization.distort({}, image=limg).image.mat, P['beq_all'])
    np.testing.assert_array_equal(D.boundary_equalization.distort({'channels': [0, 2]}, image=limg).image.mat, P['beq_c02'])
    out = D.boundary_equalization.distort({'channels': [1]}, image=limg).image
    np.testing.assert_array_equal(out.mat, P['beq_c1'])
    np.testing.assert_array_equal(D.boundary_equalization.distort({}, image=Image(mat=low[:, :, 0].copy())).image.mat,
                                  P['beq_gray'])
    np.testing.assert_array_equal(D.boundary_equalization.distort({}, image=Image(mat=big)).image.mat,
                                  O.boundary_equalization(big))
    # histogram_equalization against the oracle's cv.equalizeHist restatement
    for channels in (None, [1], [0, 2]):
        cfg = {} if channels is None else {'channels': channels}
        np.testing.assert_array_equal(D.histogram_equalization.distort(cfg, image=limg).image.mat,
                                      O.histogram_equalization(low, channels))
    np.testing.assert_array_equal(D.histogram_equalization.distort({}, image=Image(mat=big)).image.mat,
                                  O.histogram_equalization(big))
    gray = Image(mat=low[:, :, 0].copy())
    np.testing.assert_array_equal(D.histogram_equalization.distort({}, image=gray).image.mat,
                                  O.histogram_equalization(gray.mat))


def test_fog_reproduces_reference_outputs(P):
    """fog: host diamond-square field (caller's rng stream) + one page-sized alpha layer on the GPU; numpy-only in the
    reference, so the outputs below are genuine reference outputs."""
    from vkit_amd.element import Image
    from vkit_amd.mechanism import distortion as D
    img = Image(mat=P['src'])
    for i, (rough, rmax, rmin, seed) in enumerate(P['fog_cases']):
        out = D.fog.distort({'roughness': float(rough), 'ratio_max': float(rmax), 'ratio_min': float(rmin)}, image=img,
                            rng=default_rng(int(seed))).image
        np.testing.assert_array_equal(out.mat, P[f'fog_{i}'])
    wide = Image(mat=P['fog_wide_src'])
    out = D.fog.distort({'roughness': 0.6, 'fog_rgb': (200, 10, 30)}, image=wide, rng=default_rng(7)).image
    np.testing.assert_array_equal(out.mat, P['fog_wide'])
    # through an HSV image: RGB round trip around the blend, like the reference's to_rgb_image / to_original_image
    hsv = img.to_hsv_image()
    got = D.fog.distort({'roughness': 0.5}, image=hsv, rng=default_rng(0)).image
    rgb = D.fog.distort({'roughness': 0.5}, image=hsv.to_rgb_image(), rng=default_rng(0)).image
    np.testing.assert_array_equal(got.mat, O.rgb2hsv_full(rgb.mat))


def test_glass_blur_and_gather(P):
    from vkit_amd import _native as N
    from vkit_amd.element import Image
    from vkit_amd.mechanism import distortion as D
    from vkit_amd.mechanism.distortion.photometric.blur import _estimate_gaussian_kernel_size, glass_shuffle_planes
    rng = default_rng(17)
    for cn in (1, 3, 4):
        src = rng.integers(0, 256, (61, 83) + ((cn,) if cn > 1 else ()), dtype=np.uint8)
        py = rng.integers(0, 61, (40, 120))
        px = rng.integers(0, 83, (40, 120))
        np.testing.assert_array_equal(N.gather(src, py, px), src[py, px])
    with pytest.raises(N.VkxError):
        N.gather(np.zeros((4, 4), np.uint8), np.full((2, 2), 4), np.zeros((2, 2), int))
    # the operator: blur (oracle) then the reference-pinned shuffle planes (tests/test_host_golden.py)
    src = P['src']
    for sigma, delta, loop, seed in ((1.0, 1, 3, 0), (0.6, 2, 5, 1)):
        out = D.glass_blur.distort({'sigma': sigma, 'delta': delta, 'loop': loop}, image=Image(mat=src),
                                   rng=default_rng(seed)).image
        pos_y, pos_x = glass_shuffle_planes(src.shape[:2], delta, loop, default_rng(seed))
        want = O.gaussian_blur(src, _estimate_gaussian_kernel_size(sigma), sigma)[pos_y, pos_x]
        np.testing.assert_array_equal(out.mat, want)
    big = rng.integers(0, 256, (2048, 2048, 3), dtype=np.uint8)
    pos_y, pos_x = glass_shuffle_planes((2048, 2048), 1, 2, default_rng(5))
    np.testing.assert_array_equal(N.gather(big, pos_y, pos_x), big[pos_y, pos_x])


def test_poisson_noise_reproduces_reference_outputs(P):
    from vkit_amd import _native as N
    from vkit_amd.element import Image
    from vkit_amd.mechanism import distortion as D
    img = Image(mat=P['src'])
    for seed in (0, 1):
        out = D.poisson_noise.distort({}, image=img, rng=default_rng(seed)).image
        np.testing.assert_array_equal(out.mat, P[f'poisson_{seed}'])
    samples = default_rng(3).integers(-500, 900, (123, 77, 3))
    np.testing.assert_array_equal(N.saturate_i64(samples), np.clip(samples, 0, 255).astype(np.uint8))


def test_zoom_in_blur(P):
    from vkit_amd.element import Image
    from vkit_amd.mechanism import distortion as D
    img = Image(mat=P['src'])
    for i, (ratio, step, alpha) in enumerate(P['zoom_cases']):
        out = D.zoom_in_blur.distort({'ratio': float(ratio), 'step': float(step), 'alpha': float(alpha)}, image=img).image
        np.testing.assert_array_equal(out.mat, P[f'zoom_oracle_patched_{i}'])
    big = Image(mat=default_rng(23).integers(0, 256, (1024, 1024, 3), dtype=np.uint8))
    out = D.zoom_in_blur.distort({'ratio': 0.04, 'step': 0.01, 'alpha': 0.6}, image=big).image
    np.testing.assert_array_equal(out.mat, O.zoom_in_blur(big.mat, 0.04, 0.01, 0.6))


@pytest.mark.gpu
def test_resize_every_sampled_interpolation_matches_oracle():
    """cv.resize codes 0..6 (NEAREST, LINEAR, CUBIC, AREA, LANCZOS4, LINEAR_EXACT, NEAREST_EXACT) on uint8 x 1 / 3 / 4
    channels and float32 planes -- what PageResizingStep applies to the page Image, Masks and ScoreMaps.  uint8 bit-exact;
    float32 bit-exact too (same operation order, no fused multiply-add on either side)."""
    from vkit_amd import _native as N
    import oracle as O
    rng = np.random.default_rng(2024)
    shapes = [((37, 53), (20, 31)), ((37, 53), (80, 99)), ((64, 64), (32, 32)), ((60, 90), (20, 30)), ((1, 9), (4, 3)),
              ((9, 1), (3, 1)), ((50, 70), (50, 70)), ((33, 21), (32, 20)), ((128, 96), (31, 17)), ((5, 7), (11, 29))]
    for (sh, sw), (dh, dw) in shapes:
        planes = [rng.integers(0, 256, (sh, sw), dtype=np.uint8), rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8),
                  rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8), (rng.random((sh, sw)) < 0.5).astype(np.uint8),
                  rng.random((sh, sw), dtype=np.float32) * 40.0]
        for inter in range(7):
            if inter == O.INTER_AREA and (dh > sh or dw > sw):
                continue        # the reference samples INTER_AREA only when shrinking
            for src in planes:
                got = N.resize(src, (dh, dw), inter)
                want = O.resize(src, (dh, dw), inter)
                assert got.dtype == want.dtype and got.shape == want.shape
                assert (got == want).all(), (inter, src.shape, src.dtype, (dh, dw))
    with pytest.raises(N.VkxError):
        N.resize(planes[0], (sh * 2, sw * 2), O.INTER_AREA)


@pytest.mark.gpu
def test_resize_large_planes_properties():
    """Page-sized planes: identity size is a copy, constants stay constant, AREA keeps the mean."""
    from vkit_amd import _native as N
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (1024, 768, 3), dtype=np.uint8)
    mask = (rng.random((1024, 768)) < 0.3).astype(np.uint8)
    score = rng.random((1024, 768), dtype=np.float32)
    for inter in range(7):
        assert (N.resize(img, (1024, 768), inter) == img).all()
        assert (N.resize(score, (1024, 768), inter) == score).all()
        assert N.resize(mask, (700, 500), inter).shape == (700, 500)
        const = np.full((300, 200, 3), 77, np.uint8)
        assert (N.resize(const, (123, 97), inter) == 77).all()
    small = N.resize(score, (256, 192), 3)
    assert abs(float(small.mean()) - float(score.mean())) < 1e-4


@pytest.mark.gpu
def test_page_resizing_step_elements_match_oracle():
    """PageResizingStep.run on real elements: every output equals the oracle's cv.resize restatement applied the way
    the reference's element methods do (mask x 255 -> resize -> > 0; score map resize then x ratio)."""
    from types import SimpleNamespace
    import oracle as O
    from vkit_amd.element import Image, Mask, ScoreMap
    from vkit_amd.pipeline.text_detection import PageResizingStep, PageResizingStepConfig, PageResizingStepInput
    from vkit_amd.utility import sample_cv_resize_interpolation
    rng = np.random.default_rng(11)
    h, w = 240, 320
    used = set()
    for seed in range(12):
        page = SimpleNamespace(
            page_image=Image(mat=rng.integers(0, 256, (h, w, 3), dtype=np.uint8)),
            page_active_mask=Mask(mat=(rng.random((h, w)) < 0.9).astype(np.uint8)),
            page_char_mask=Mask(mat=(rng.random((h, w)) < 0.2).astype(np.uint8)),
            page_seal_impression_char_mask=Mask(mat=(rng.random((h, w)) < 0.05).astype(np.uint8)),
            page_char_height_score_map=ScoreMap(mat=(rng.random((h, w), dtype=np.float32) * 30).astype(np.float32), is_prob=False),
            page_text_line_mask=Mask(mat=(rng.random((h, w)) < 0.4).astype(np.uint8)),
            page_text_line_height_score_map=ScoreMap(mat=(rng.random((h, w), dtype=np.float32) * 30).astype(np.float32), is_prob=False),
            page_text_line_heights=[float(v) for v in rng.uniform(4.0, 30.0, 12)])
        step = PageResizingStep(PageResizingStepConfig(resized_text_line_height_min=3.0, resized_text_line_height_max=24.0))
        out = step.run(PageResizingStepInput(page_distortion_step_output=page), np.random.default_rng(seed))
        # the same page with its elements resident on the device (what PageDistortionStep hands over: the four masks then go through
        # two batched look-ups around their resizes) has to give the same planes
        from vkit_amd import _native
        ctx = _native.default_ctx()
        resident = SimpleNamespace(page_text_line_heights=page.page_text_line_heights)
        for name in ('page_image', 'page_active_mask', 'page_char_mask', 'page_seal_impression_char_mask', 'page_text_line_mask'):
            element = getattr(page, name)
            setattr(resident, name, type(element)(mat=ctx.to_device(element.mat)))
        for name in ('page_char_height_score_map', 'page_text_line_height_score_map'):
            setattr(resident, name, ScoreMap(mat=ctx.to_device(getattr(page, name).mat), is_prob=False))
        out_dev = step.run(PageResizingStepInput(page_distortion_step_output=resident), np.random.default_rng(seed))
        for name in ('page_image', 'page_active_mask', 'page_char_mask', 'page_seal_impression_char_mask', 'page_text_line_mask',
                     'page_char_height_score_map', 'page_text_line_height_score_map'):
            assert (getattr(out_dev, name).mat == getattr(out, name).mat).all(), (name, 'device-resident')
        # replay the two draws
        r = np.random.default_rng(seed)
        ratio = r.uniform(3.0, 24.0) / step.get_text_line_heights_min(page.page_text_line_heights)
        size = (round(ratio * h), round(ratio * w))
        inter = sample_cv_resize_interpolation(r, include_cv_inter_area=(ratio < 1.0))
        used.add(inter)
        assert out.page_image.shape == size
        assert (out.page_image.mat == O.resize(page.page_image.mat, size, inter)).all()
        for name in ('page_active_mask', 'page_char_mask', 'page_seal_impression_char_mask', 'page_text_line_mask'):
            want = (O.resize(getattr(page, name).mat * 255, size, inter) > 0).astype(np.uint8)
            assert (getattr(out, name).mat == want).all(), (name, inter)
        for name in ('page_char_height_score_map', 'page_text_line_height_score_map'):
            want = O.resize(getattr(page, name).mat, size, inter) * ratio
            assert (getattr(out, name).mat == want).all(), (name, inter)
    assert len(used) >= 4


@pytest.mark.gpu
def test_std_shift_matches_reference_goldens(golden_dir):
    """std_shift end to end (host numpy mean, device table pass) against outputs of the reference itself."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_golden import _std_shift_cases
    import vkit_amd.mechanism.distortion as D
    from vkit_amd.element import Image
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


@pytest.mark.gpu
def test_filter2d_matches_oracle():
    """vkx_filter2d_u8 against the restatement of cv.filter2D: odd and even kernel sizes up to 15 x 15, zero taps,
    negative taps (saturation both ways), 1 / 3 / 4 channels, images smaller than the kernel, ragged sizes."""
    from vkit_amd import _native as N
    import oracle as O
    rng = np.random.default_rng(42)
    for shape in [(37, 53, 3), (64, 64), (1, 9, 3), (9, 1), (130, 70, 4), (3, 3, 3), (200, 333, 3)]:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        for kh, kw in [(1, 1), (3, 3), (5, 5), (7, 7), (15, 15), (3, 7), (4, 4), (2, 5)]:
            kernel = rng.random((kh, kw)).astype(np.float32)
            kernel[rng.random((kh, kw)) < 0.3] = 0
            kernel /= max(kernel.sum(), np.float32(1e-3))
            if (kh + kw) % 3 == 0:
                kernel = (kernel * 3 - np.float32(0.2)).astype(np.float32)       # overshoot and negatives
            got = N.filter2d(img, kernel)
            assert (got == O.filter2d(img, kernel)).all(), (shape, kh, kw)
    with pytest.raises(N.VkxError):
        N.filter2d(img, np.ones((16, 16), np.float32))


@pytest.mark.gpu
def test_defocus_and_motion_blur_match_oracle():
    """defocus_blur / motion_blur end to end (host kernel construction incl. the device warpAffine of the motion line,
    device filter2D) against the oracle pipeline, for every radius / a sweep of angles, RGB and grayscale."""
    from vkit_amd import _native as N
    import oracle as O
    import vkit_amd.mechanism.distortion as D
    from vkit_amd.element import Image
    from vkit_amd.mechanism.distortion.photometric import blur as B
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, (77, 41), dtype=np.uint8)
    for radius in (1, 2, 3):
        for mat in (rgb, gray):
            got = D.defocus_blur.distort(D.DefocusBlurConfig(radius=radius), image=Image(mat=mat)).image.mat
            assert (got == O.defocus_blur(mat, radius)).all()
        for angle in (0, 17, 45, 90, 133, 180, 271, 306, 359, 360, 725):
            k = B.motion_kernel(radius, angle, 0.5)
            assert (k == O.motion_kernel(radius, angle, 0.5)).all(), (radius, angle)
            got = D.motion_blur.distort(D.MotionBlurConfig(radius=radius, angle=angle), image=Image(mat=rgb)).image.mat
            assert (got == O.motion_blur(rgb, radius, angle)).all(), (radius, angle)
    # a horizontal line at angle 0 only blurs along x (plus the anti-aliasing spread)
    k0 = B.motion_kernel(2, 0, 0.5)
    assert k0[3].sum() > 0.75 * k0.sum()


@pytest.mark.gpu
def test_resize_table_cache_survives_eviction():
    """The context keeps the tap tables of the last few resize geometries: cycling through more geometries than slots,
    interleaving interpolations and dtypes, must keep returning the oracle's planes (stale or mixed-up tables would not)."""
    from vkit_amd import _native as N
    import oracle as O
    rng = np.random.default_rng(99)
    img = rng.integers(0, 256, (61, 83, 3), dtype=np.uint8)
    plane = rng.random((61, 83), dtype=np.float32)
    sizes = [(40, 50), (90, 120), (61, 84), (30, 83), (122, 166), (45, 45), (100, 30), (64, 64), (40, 50), (90, 120)]
    for rep in range(3):
        for dh, dw in sizes:
            for inter in (2, 4, 5, 3):
                if inter == 3 and (dh > 61 or dw > 83):
                    continue
                assert (N.resize(img, (dh, dw), inter) == O.resize(img, (dh, dw), inter)).all(), (rep, dh, dw, inter)
                assert (N.resize(plane, (dh, dw), inter) == O.resize(plane, (dh, dw), inter)).all(), (rep, dh, dw, inter)
    # a different source with a cached geometry must not reuse anything but the tables
    other = rng.integers(0, 256, (61, 83, 3), dtype=np.uint8)
    assert (N.resize(other, (90, 120), 4) == O.resize(other, (90, 120), 4)).all()


@pytest.mark.gpu
def test_resize_lanczos4_nan_coefficient_column():
    """Found by tools/soak.py: when the float32 fraction of a destination index rounds up to exactly 1.0 (e.g. 4 -> 196
    columns, index 24), interpolateLanczos4 divides 0 by 0 for one tap and the whole coefficient vector becomes
    (0, 0, 0, 0, NaN, 0, 0, 0).  cv2 turns the NaN into -32768 through saturate_cast<short>(cvRound(NaN)) on uint8 and
    carries it on float32; both must come out of the GPU path the same way."""
    from vkit_amd import _native as N
    import oracle as O
    rng = np.random.default_rng(0)
    for (sh, sw), (dh, dw) in (((146, 4), (64, 196)), ((127, 1), (202, 197)), ((4, 146), (196, 64))):
        for cn in (1, 3, 4):
            src = rng.integers(0, 256, (sh, sw) if cn == 1 else (sh, sw, cn), dtype=np.uint8)
            assert (N.resize(src, (dh, dw), 4) == O.resize(src, (dh, dw), 4)).all()
        plane = rng.random((sh, sw), dtype=np.float32)
        got, want = N.resize(plane, (dh, dw), 4), O.resize(plane, (dh, dw), 4)
        assert np.isnan(want).any()
        assert ((got == want) | (np.isnan(got) & np.isnan(want))).all()


def test_throughput_noise_plane_matches_its_definition():
    """vkx_noise_normal_i16 (device Philox2x32-10 + inverse-CDF table) against the oracle's statement of the same
    definition, bit for bit; the chain with a device-drawn plane equals the chain fed the same plane from the host."""
    from vkit_amd import _native as N
    for shape, std, seed in (((123, 77, 3), 10.0, 1), ((64, 64, 1), 3.0, 2 ** 63 + 5), ((31, 500, 4), 25.0, 0xdeadbeefcafe),
                             ((2147, 2115, 3), 10.0, 42), ((50, 70, 3), 100.0, 9), ((3, 5, 1), 0.3, 4)):
        got = N.noise_normal_i16(shape, std, seed)
        want = O.noise_normal_i16(shape, std, seed)
        assert got.dtype == np.int16 and got.shape == want.shape and (got == want).all(), shape
    assert (N.noise_normal_table(7.5) == O.noise_normal_table(7.5)).all()
    with pytest.raises(N.VkxError):
        N.noise_normal_i16((4, 4, 3), 0.0, 1)
    # the planes of a batch in one launch (vkx_noise_normal_i16_batch_dev): ragged shapes, own seeds, an empty plane, and
    # more planes than the kernel-argument form holds
    ctx = N.default_ctx()
    shapes = [(123, 77, 3), (1, 1, 1), (0, 5, 3), (301, 257, 3), (64, 64, 4)] + [(17 + k, 29, 3) for k in range(9)]
    seeds = [0x1234 + 977 * k for k in range(len(shapes))]
    outs = [ctx.dev_empty(shape if shape[0] else (1, 1, 1), np.int16) for shape in shapes]
    planes = (N.VkxNoisePlane * len(shapes))()
    for pl, out, shape, seed in zip(planes, outs, shapes, seeds):
        pl.dst, pl.stride_el, pl.h, pl.w, pl.cn, pl.seed = out.ptr, shape[1] * shape[2], shape[0], shape[1], shape[2], seed
    N.check(N.lib().vkx_noise_normal_i16_batch_dev(ctx.handle, planes, len(shapes), 12.5))
    for out, shape, seed in zip(outs, shapes, seeds):
        if shape[0]:
            assert (out.host() == O.noise_normal_i16(shape, 12.5, seed)).all(), shape


def test_dense_and_pitched_planes_agree():
    """The element-wise and RGB pixel kernels have two bodies: 16 bytes (4 pixels) per lane on dense, aligned planes, one
    byte (pixel) per lane on any pitch.  Both must produce the oracle's bytes: every entry point is run on a dense plane, on
    a padded one (odd pitch: unaligned rows) and, where the operator allows it, in place."""
    import ctypes
    from vkit_amd import _native as N
    ctx, lib = N.default_ctx(), N.lib()
    rng = default_rng(77)

    def run(call, src, cn, pitch_pad, extra=None, inplace=False):
        """call(src_ptr, src_pitch, dst_ptr, dst_pitch, extra_ptr, extra_pitch_el) on device copies; returns the result"""
        h, w = src.shape[:2]
        row = w * cn
        pitch = row + pitch_pad
        host = np.zeros((h, pitch), np.uint8)
        host[:, :row] = src.reshape(h, row)
        d_src, d_dst = ctx.malloc(host.nbytes + 64), ctx.malloc(host.nbytes + 64)
        ctx.upload(d_src, host)
        ctx.upload(d_dst, np.full((h, pitch), 0xA5, np.uint8))
        d_extra, extra_pitch = 0, 0
        if extra is not None:
            per = extra.size // h
            extra_pitch = per + (pitch_pad if pitch_pad else 0)
            e = np.zeros((h, extra_pitch), extra.dtype)
            e[:, :per] = extra.reshape(h, per)
            d_extra = ctx.malloc(e.nbytes + 64)
            ctx.upload(d_extra, e)
        N.check(call(d_src, pitch, d_src if inplace else d_dst, pitch, d_extra, extra_pitch))
        out = np.zeros((h, pitch), np.uint8)
        ctx.download(d_src if inplace else d_dst, out)
        ctx.sync()
        for p in (d_src, d_dst, d_extra):
            if p:
                ctx.free(p)
        if not inplace:
            assert (out[:, row:] == 0xA5).all()          # nothing written into the padding
        return out[:, :row].reshape(src.shape)

    for (h, w) in ((37, 53), (64, 64), (1, 7), (5, 1), (130, 257)):
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        gray = rng.integers(0, 256, (h, w), dtype=np.uint8)
        rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        noise = rng.integers(-300, 300, (h, w, 3)).astype(np.int16)
        noise[0, 0] = (32767, -32768, 255)
        sel = rng.integers(0, 3, (h, w)).astype(np.uint8)
        cases = []
        for delta in (37, -200, 0):
            cases.append((f'color_shift {delta}', rgb, 3, None, O.color_shift_rgb(rgb, delta), True,
                          lambda s, sp, d, dp, e, ep, delta=delta: lib.vkx_color_shift_rgb_dev(ctx.handle, s, h, w, sp, delta, d, dp)))
        cases.append(('rgb2hsv', rgb, 3, None, None, True,
                      lambda s, sp, d, dp, e, ep: lib.vkx_cvt_rgb_hsv_u8_dev(ctx.handle, s, h, w, sp, 1, d, dp)))
        cases.append(('hsv2rgb', rgb, 3, None, None, True,
                      lambda s, sp, d, dp, e, ep: lib.vkx_cvt_rgb_hsv_u8_dev(ctx.handle, s, h, w, sp, 0, d, dp)))
        cases.append(('brightness', rgb, 3, None, O.brightness_shift_rgb(rgb, 20), True,
                      lambda s, sp, d, dp, e, ep: lib.vkx_brightness_shift_rgb_dev(ctx.handle, s, h, w, sp, 20, d, dp)))
        cases.append(('color_balance', rgb, 3, None, O.color_balance_rgb(rgb, 0.4), True,
                      lambda s, sp, d, dp, e, ep: lib.vkx_color_balance_rgb_dev(ctx.handle, s, h, w, sp, ctypes.c_double(0.4), d, dp)))
        for mat, cn in ((rgb, 3), (gray, 1), (rgba, 4)):
            cases.append((f'mean_shift cn{cn}', mat, cn, None, None, True,
                          lambda s, sp, d, dp, e, ep, cn=cn: lib.vkx_mean_shift_u8_dev(ctx.handle, s, h, w, cn, sp, 40, 1, 128, 0, 0b101, d, dp)))
            cases.append((f'mean_shift cycle cn{cn}', mat, cn, None, None, True,
                          lambda s, sp, d, dp, e, ep, cn=cn: lib.vkx_mean_shift_u8_dev(ctx.handle, s, h, w, cn, sp, -77, 0, 0, 1, 0, d, dp)))
            cases.append((f'complement cn{cn}', mat, cn, None, None, True,
                          lambda s, sp, d, dp, e, ep, cn=cn: lib.vkx_pointwise_u8_dev(ctx.handle, s, h, w, cn, sp, 0, 100, 1, 0b011, d, dp)))
            cases.append((f'posterize cn{cn}', mat, cn, None, None, True,
                          lambda s, sp, d, dp, e, ep, cn=cn: lib.vkx_pointwise_u8_dev(ctx.handle, s, h, w, cn, sp, 1, 5, 0, 0, d, dp)))
            cases.append((f'impulse cn{cn}', mat, cn, sel, None, True,
                          lambda s, sp, d, dp, e, ep, cn=cn: lib.vkx_impulse_noise_u8_dev(ctx.handle, s, h, w, cn, sp, e, ep, d, dp)))
            lut = rng.integers(0, 256, (cn, 256), dtype=np.uint8)
            cases.append((f'lut cn{cn}', mat, cn, None, None, True,
                          lambda s, sp, d, dp, e, ep, cn=cn, lut=lut: lib.vkx_apply_lut_u8_dev(ctx.handle, s, h, w, cn, sp, lut.ctypes.data, 0b110, d, dp)))
        cases.append(('add_noise', rgb, 3, noise, O.add_noise_i16(rgb, noise), True,
                      lambda s, sp, d, dp, e, ep: lib.vkx_add_noise_i16_dev(ctx.handle, s, h, w, 3, sp, e, ep, d, dp)))
        for name, mat, cn, extra, want, can_inplace, call in cases:
            dense = run(call, mat, cn, 0, extra)
            pitched = run(call, mat, cn, 5, extra)
            np.testing.assert_array_equal(dense, pitched, err_msg=f'{name} {(h, w)}')
            if want is not None:
                np.testing.assert_array_equal(dense, want, err_msg=f'{name} {(h, w)} vs oracle')
            if can_inplace:
                np.testing.assert_array_equal(run(call, mat, cn, 0, extra, inplace=True), dense, err_msg=f'{name} in place')


def test_histogram_dense_and_pitched():
    from vkit_amd import _native as N
    ctx, lib = N.default_ctx(), N.lib()
    rng = default_rng(78)
    for (h, w, cn) in ((37, 53, 3), (64, 64, 1), (1, 7, 4), (300, 517, 3), (2048, 2048, 3)):
        mat = rng.integers(0, 256, (h, w, cn), dtype=np.uint8)
        mat[: h // 2] //= 3                                           # crowded bins
        want = np.stack([np.bincount(mat[:, :, c].ravel(), minlength=256) for c in range(cn)]).astype(np.int32)
        for pad in (0, 3):
            pitch = w * cn + pad
            host = np.zeros((h, pitch), np.uint8)
            host[:, :w * cn] = mat.reshape(h, -1)
            d_src, d_hist = ctx.malloc(host.nbytes + 64), ctx.malloc(4 * 256 * 4)
            ctx.upload(d_src, host)
            N.check(lib.vkx_histogram_u8_dev(ctx.handle, d_src, h, w, cn, pitch, d_hist))
            got = np.zeros((cn, 256), np.int32)
            ctx.download(d_hist, got)
            ctx.sync()
            ctx.free(d_src); ctx.free(d_hist)
            np.testing.assert_array_equal(got, want, err_msg=str((h, w, cn, pad)))
