"""The order numpy adds in when ``std_shift`` takes ``np.mean`` of a float32 image (photometric/color.py:165-210), restated in
integers and checked against numpy itself: the statement csrc/reduce.hip implements on the device.

* one plane, and channels picked with ``mat[:, :, channels]`` (numpy lays that copy out channel first, so the reduced axis is the
  contiguous one): per channel, pieces of 8 192 contiguous elements (numpy's reduction buffer), each summed exactly (a pairwise
  sum of at most 8 192 x 255 stays below 2^24), accumulated one after the other in float32;
* all C >= 2 channels of an interleaved image, ``axis=0`` of ``(N, C)``: a sequential float32 accumulation over the pixels.  In the binade [2^(23+e), 2^(24+e))
  every addition rounds the addend to a multiple of q = 2^e, a tie to the even multiple: (parity of s / q) is the only state."""
import numpy as np
import pytest
from numpy.random import default_rng

f32 = np.float32


def pieces_sum(plane_u8):
    flat = plane_u8.reshape(-1).astype(np.int64)
    s = f32(0)
    for lo in range(0, flat.size, 8192):
        s = f32(s + f32(int(flat[lo:lo + 8192].sum())))
    return s


def sequential_sum_by_binades(col_u8, run=256):
    """The device algorithm in Python integers: per run and binade the (units, parity) transfer, a walk over the runs."""
    v = col_u8.astype(np.int64)
    s = 0
    for lo in range(0, v.size, run):
        seg = v[lo:lo + run]
        bl = max(s, 1).bit_length()
        e = max(0, bl - 24)
        if e == 0:
            nxt = s + int(seg.sum())
            ok = nxt < (1 << 24)
        else:
            q, half = 1 << e, 1 << (e - 1)
            par, units = (s >> e) & 1, 0
            for x in seg:
                up, b = int(x) >> e, int(x) & (q - 1)
                t = up + (1 if b > half else 0) + (1 if (b == half and ((par + up) & 1)) else 0)
                units += t
                par = (par + t) & 1
            nxt = s + (units << e)
            ok = nxt < (1 << bl)
        if not ok:
            f = f32(s)
            for x in seg:
                f = f32(f + f32(x))
            nxt = int(f)
        s = nxt
    return f32(s)


@pytest.mark.parametrize('shape,hi', [((37, 53), 256), ((300, 411), 256), ((1024, 1000), 256), ((700, 900), 1), ((1100, 1300), 255)])
def test_single_plane_mean_is_pieces_of_8192(shape, hi):
    rng = default_rng(sum(shape))
    img = rng.integers(0, 256, shape, dtype=np.uint8) if hi == 256 else np.full(shape, hi, np.uint8)
    want = np.mean(img.astype(np.float32))
    got = pieces_sum(img) / img.size
    assert want.dtype == np.float32 and f32(got) == want
    # one selected channel of a colour image: the fancy-indexed copy reduces the same way
    rgb = np.stack([img, img[::-1], img[:, ::-1]], axis=-1)
    mat = rgb[:, :, [1]].astype(np.float32)
    want1 = np.mean(mat.reshape(-1, 1), axis=0)
    assert f32(pieces_sum(rgb[:, :, 1]) / img.size) == want1[0]


@pytest.mark.parametrize('n,mode', [(1999, 'random'), (70_000, 'random'), (300_000, 'random'), (400_000, 'all255'), (600_000, 'odd'),
                                    (1 << 20, 'random'), (1 << 20, 'all255')])
def test_colour_mean_is_a_sequential_float32_sum(n, mode):
    rng = default_rng(n)
    if mode == 'random':
        px = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    elif mode == 'all255':
        px = np.full((n, 3), 255, np.uint8)
    else:
        px = (rng.integers(0, 128, (n, 3)) * 2 + 1).astype(np.uint8)          # ties in every binade
    want = np.mean(px.astype(np.float32).reshape(-1, 3), axis=0)
    for c in range(3):
        assert f32(sequential_sum_by_binades(px[:, c]) / n) == want[c], c


def test_picked_channels_reduce_piecewise():
    """``mat[:, :, channels]`` is an advanced index: the copy is channel-first in memory, ``reshape(-1, k)`` a view of it, and the
    reduction of every channel runs over contiguous elements -- pieces of 8 192, not the sequential form of the interleaved case."""
    rng = default_rng(5)
    img = rng.integers(0, 256, (1300, 1100, 4), dtype=np.uint8)
    for channels in ([3, 1, 0], [2], [0, 1, 2, 3]):
        mat = img[:, :, channels].astype(np.float32)
        want = np.mean(mat.reshape(-1, mat.shape[-1]), axis=0)
        for k, c in enumerate(channels):
            assert f32(pieces_sum(img[:, :, c]) / (1300 * 1100)) == want[k], (channels, c)


def test_device_mean_is_gated_on_numpy_adding_in_the_restated_order():
    """``_native.numpy_reduce_order_ok``: true on the numpy the kernel was written against, false as soon as the reduction buffer
    is not the default one (std_shift then takes np.mean itself)."""
    from vkit_amd import _native as N
    assert N.numpy_reduce_order_ok()
    old = np.getbufsize()
    try:
        np.setbufsize(16384)
        assert not N.numpy_reduce_order_ok()
    finally:
        np.setbufsize(old)
    assert N.numpy_reduce_order_ok()
