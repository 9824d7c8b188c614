"""Image modes beyond RGB through the photometric operators: RGBA and GRAYSCALE conversions, color_shift / color_balance on
every mode, fog on GRAYSCALE (reference: vkit/element/image.py:188-216, 771-814; photometric/color.py:93-116, 371-396;
photometric/effect.py:170-209).  The expected values are the reference's numpy expressions over the oracle's colour
conversions."""
import numpy as np
import pytest

import oracle as O
from vkit_amd import _native as N
from vkit_amd.element import Image, ImageMode
from vkit_amd.mechanism.distortion import color_balance, color_shift, fog
from vkit_amd.mechanism.distortion.photometric.color import ColorBalanceConfig, ColorShiftConfig
from vkit_amd.mechanism.distortion.photometric.effect import FogConfig, generate_diamond_square_mask

pytestmark = pytest.mark.gpu


def _rgba(seed, h=97, w=131):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 4), dtype=np.uint8)


def test_alpha_conversions():
    rgba = _rgba(0)
    rgb, gray = rgba[:, :, :3].copy(), O.rgb2gray(rgba[:, :, :3].copy())
    assert (N.cvt_color(rgba, N.CVT_RGBA2RGB) == rgb).all()
    assert (N.cvt_color(rgba, N.CVT_RGBA2GRAY) == gray).all()
    back = N.cvt_color(rgb, N.CVT_RGB2RGBA)
    assert (back[:, :, :3] == rgb).all() and (back[:, :, 3] == 255).all()
    g4 = N.cvt_color(gray, N.CVT_GRAY2RGBA)
    assert all((g4[:, :, c] == gray).all() for c in range(3)) and (g4[:, :, 3] == 255).all()
    image = Image(mat=rgba)
    assert image.mode == ImageMode.RGBA
    assert (image.to_rgb_image().mat == rgb).all()
    assert (image.to_grayscale_image().mat == gray).all()
    assert (image.to_hsv_image().mat == O.rgb2hsv_full(rgb)).all()
    hsl = image.to_hsl_image().mat
    assert (hsl == O.rgb2hls_full(rgb)[:, :, [0, 2, 1]]).all()
    assert (Image(mat=gray).to_rgba_image().mat == g4).all()
    assert (Image(mat=rgb).to_rgba_image().mat == back).all()


def _hue_add(hsv, delta):
    out = hsv.copy()
    out[:, :, 0] = ((hsv[:, :, 0].astype(np.int16) + delta) % 256).astype(np.uint8)
    return out


@pytest.mark.parametrize('delta', [37, -90, 0])
def test_color_shift_on_every_mode(delta):
    rgba = _rgba(1)
    rgb = rgba[:, :, :3].copy()
    # RGBA: to HSV (alpha dropped), hue add, back to RGBA with alpha 255
    want_rgb = O.hsv2rgb_full(_hue_add(O.rgb2hsv_full(rgb), delta))
    got = color_shift.distort(ColorShiftConfig(delta=delta), image=Image(mat=rgba)).image
    assert got.mode == ImageMode.RGBA
    assert (got.mat[:, :, :3] == want_rgb).all() and (got.mat[:, :, 3] == 255).all()
    # GRAYSCALE: grey -> RGB -> HSV (S = 0) -> hue add -> RGB -> grey: the grey plane itself
    gray = O.rgb2gray(rgb)
    got = color_shift.distort(ColorShiftConfig(delta=delta), image=Image(mat=gray)).image
    assert got.mode == ImageMode.GRAYSCALE
    rgb3 = np.repeat(gray[:, :, None], 3, axis=2)
    want = O.rgb2gray(O.hsv2rgb_full(_hue_add(O.rgb2hsv_full(rgb3), delta)))
    assert (got.mat == want).all()
    # HSV / HSL images only take the hue add
    hsv = O.rgb2hsv_full(rgb)
    got = color_shift.distort(ColorShiftConfig(delta=delta), image=Image(mat=hsv, mode=ImageMode.HSV)).image
    assert (got.mat == _hue_add(hsv, delta)).all()


def _balance(gray_like, mat, ratio, channels=None):
    g, m = gray_like.astype(np.float32), mat.astype(np.float32)
    if channels:
        g, m = g[:, :, channels], m[:, :, channels]
    out = np.clip((1 - ratio) * g + ratio * m, 0, 255).astype(np.uint8)
    if channels:
        full = mat.copy()
        full[:, :, channels] = out
        return full
    return out


@pytest.mark.parametrize('ratio', [0.0, 0.3, 0.85, 1.0])
def test_color_balance_on_every_mode(ratio):
    rgba = _rgba(2)
    rgb = rgba[:, :, :3].copy()
    gray = O.rgb2gray(rgb)
    rgb3 = np.repeat(gray[:, :, None], 3, axis=2)
    # RGBA: every channel, alpha included, against (g, g, g, 255)
    g4 = np.concatenate([rgb3, np.full(gray.shape + (1,), 255, np.uint8)], axis=2)
    got = color_balance.distort(ColorBalanceConfig(ratio=ratio), image=Image(mat=rgba)).image
    assert (got.mat == _balance(g4, rgba, ratio)).all()
    # HSV / HSL: channels 1 and 2 against the grey image brought to that mode
    hsv = O.rgb2hsv_full(rgb)
    gray_hsv = O.rgb2hsv_full(np.repeat(O.rgb2gray(O.hsv2rgb_full(hsv))[:, :, None], 3, axis=2))
    got = color_balance.distort(ColorBalanceConfig(ratio=ratio), image=Image(mat=hsv, mode=ImageMode.HSV)).image
    assert (got.mat == _balance(gray_hsv, hsv, ratio, [1, 2])).all()
    hsl = O.rgb2hls_full(rgb)[:, :, [0, 2, 1]].copy()
    rgb_from_hsl = O.hls2rgb_full(hsl[:, :, [0, 2, 1]].copy())
    gray_hsl = O.rgb2hls_full(np.repeat(O.rgb2gray(rgb_from_hsl)[:, :, None], 3, axis=2))[:, :, [0, 2, 1]]
    got = color_balance.distort(ColorBalanceConfig(ratio=ratio), image=Image(mat=hsl, mode=ImageMode.HSL)).image
    assert (got.mat == _balance(gray_hsl, hsl, ratio, [1, 2])).all()
    # GRAYSCALE comes back as it is
    image = Image(mat=gray)
    assert color_balance.distort(ColorBalanceConfig(ratio=ratio), image=image).image is image


def test_fog_on_grayscale_and_rgba():
    gray = np.random.default_rng(3).integers(0, 256, (120, 90), dtype=np.uint8)
    config = FogConfig(roughness=0.6, ratio_max=0.9, ratio_min=0.1)
    got = fog.distort(config, image=Image(mat=gray), rng=np.random.default_rng(7)).image

    def field(shape):
        mask = np.array(generate_diamond_square_mask(shape, config.roughness, np.random.default_rng(7)), dtype=np.float32)
        mask -= mask.min()
        mask /= mask.max()
        mask *= (config.ratio_max - config.ratio_min)
        mask += config.ratio_min
        return mask

    mask = field(gray.shape)
    val = 0.2126 * config.fog_rgb[0] + 0.7152 * config.fog_rgb[1] + 0.0722 * config.fog_rgb[2]
    want = np.clip((1 - mask) * gray.astype(np.float32) + mask * np.full(gray.shape, val, dtype=np.float32), 0, 255).astype(np.uint8)
    assert got.mode == ImageMode.GRAYSCALE and (got.mat == want).all()

    rgba = _rgba(4, 120, 90)
    got = fog.distort(config, image=Image(mat=rgba), rng=np.random.default_rng(7)).image
    m3 = np.expand_dims(field(rgba.shape[:2]), -1)
    rgb = rgba[:, :, :3].astype(np.float32)
    want_rgb = np.clip((1 - m3) * rgb + m3 * np.full(rgb.shape, config.fog_rgb, dtype=np.float32), 0, 255).astype(np.uint8)
    assert got.mode == ImageMode.RGBA and (got.mat[:, :, :3] == want_rgb).all() and (got.mat[:, :, 3] == 255).all()
