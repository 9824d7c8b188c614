"""std_shift's mean on the device (csrc/reduce.hip: vkx_sum_f32_u8) against numpy ITSELF: np.mean of the float32 copy, per selected
channel, value for value -- including the roundings of running sums beyond 2^24.  Reference: photometric/color.py:165-210."""
import numpy as np
import pytest
from numpy.random import default_rng

from vkit_amd import _native as N
from vkit_amd.element import Image
from vkit_amd.mechanism import distortion as D

pytestmark = pytest.mark.gpu


def _numpy_mean(img, channels):
    mat = img[:, :, list(channels)] if channels else img
    mat = mat.astype(np.float32)
    return np.mean(mat) if mat.ndim == 2 else np.mean(mat.reshape(-1, mat.shape[-1]), axis=0)


@pytest.mark.parametrize('shape,channels,fill', [
    ((1, 1, 3), None, None), ((37, 53, 3), None, None), ((37, 53, 3), [2, 0], None), ((64, 48), None, None), ((300, 411, 3), [1], None),
    ((1024, 1024, 3), None, None), ((1024, 1024, 3), None, 255), ((1300, 1100, 4), [3, 1, 0], None), ((2048, 2048, 3), None, 255),
    ((2048, 2048, 3), None, None), ((2048, 2048), None, None), ((2048, 2048), None, 255), ((1500, 999, 3), None, 'odd'),
])
def test_device_mean_is_numpys(shape, channels, fill):
    rng = default_rng(len(shape) * 1000 + shape[0])
    if fill is None:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
    elif fill == 'odd':
        img = (rng.integers(0, 128, shape) * 2 + 1).astype(np.uint8)
    else:
        img = np.full(shape, fill, np.uint8)
    want = _numpy_mean(img, channels)
    got = N.mean_f32_u8(img, channels if img.ndim == 3 else None)
    assert np.asarray(got).dtype == np.float32
    assert (np.asarray(got) == np.asarray(want)).all(), (got, want)
    dev = N.mean_f32_u8(N.default_ctx().to_device(img), channels if img.ndim == 3 else None)       # device-resident input
    assert (np.asarray(dev) == np.asarray(want)).all()


def test_std_shift_on_a_large_page_equals_the_reference_formula(monkeypatch):
    """The operator end to end at page size: device mean + table pass == the reference's float32 expression evaluated by numpy."""
    rng = default_rng(7)
    for shape, channels, scale in (((1024, 1280, 3), None, 1.37), ((1400, 1100, 3), [0, 2], 0.6), ((900, 1200), None, 1.9)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        got = D.std_shift.distort(D.StdShiftConfig(scale=scale, channels=channels), image=Image(mat=img)).image.mat
        mat = (img[:, :, channels] if channels else img).astype(np.float32)
        mean = np.mean(mat) if mat.ndim == 2 else np.mean(mat.reshape(-1, mat.shape[-1]), axis=0)
        out = np.clip(np.round(mat * scale - mean * (scale - 1)), 0, 255).astype(np.uint8)
        want = img.copy()
        if channels:
            want[:, :, channels] = out
        else:
            want = out
        assert (got == want).all(), (shape, channels)
        monkeypatch.setenv('VKX_HOST_MEAN', '1')            # the numpy fallback gives the same pixels
        assert (D.std_shift.distort(D.StdShiftConfig(scale=scale, channels=channels), image=Image(mat=img)).image.mat == want).all()
        monkeypatch.delenv('VKX_HOST_MEAN')
