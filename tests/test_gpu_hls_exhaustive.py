"""Every 8-bit colour through brightness_shift (RGB -> HLS, L + delta, HLS -> RGB; photometric/color.py) and the two conversions on
their own, against the oracle's restatement of OpenCV's float HLS formulas with plain IEEE divisions: the device takes both quotients
through a reciprocal + correction sequence (photo.hip: div_normal) that is exact only for normal operands -- which all 2^24 colours
are checked to be."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _all_colours():
    v = np.arange(1 << 24, dtype=np.uint32)
    return np.stack([v & 255, (v >> 8) & 255, v >> 16], axis=-1).astype(np.uint8).reshape(4096, 4096, 3)


def test_all_colours_brightness_and_hls():
    import oracle as O
    from vkit_amd import _native as N
    img = _all_colours()
    for delta in (0, 37, -90):
        np.testing.assert_array_equal(N.brightness_shift_rgb(img, delta), O.brightness_shift_rgb(img, delta))
    np.testing.assert_array_equal(N.cvt_color(img, N.CVT_RGB2HLS_FULL), O.rgb2hls_full(img))
    np.testing.assert_array_equal(N.cvt_color(img, N.CVT_HLS2RGB_FULL), O.hls2rgb_full(img))
