"""Edge cases through the C ABI: empty inputs, 1-pixel planes, the 32767-px limit, and planes larger than 2^31 bytes of
address arithmetic checked through size-independent properties (no oracle at that size)."""
import numpy as np
import pytest
from numpy.random import default_rng

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def N():
    from vkit_amd import _native
    return _native


def test_empty_inputs(N):
    from vkit_amd.batch import ChainBatch
    page = default_rng(0).integers(0, 256, (8, 9, 3), dtype=np.uint8)
    before = page.copy()
    N.fill(page, [])                                            # no layers
    np.testing.assert_array_equal(page, before)
    layer = N.make_layer((2, 3, 0, 4), 3, (1, 2, 3))            # zero-height box
    N.fill(page, [layer, layer])
    np.testing.assert_array_equal(page, before)
    mask = np.zeros((8, 9), np.uint8)
    N.paint_polys([], mask=mask)
    assert not mask.any()
    batch = ChainBatch()
    batch.run()                                                 # zero images
    batch.close()
    for shape in ((0, 5, 3), (5, 0, 3)):
        empty = np.zeros(shape, np.uint8)
        assert N.mean_shift(empty, 10).shape == shape
        assert N.pointwise(empty, N.POINT_COMPLEMENT, -1).shape == shape
        assert N.add_noise_i16(empty, np.zeros(shape, np.int16)).shape == shape


def test_one_pixel_planes(N):
    rng = default_rng(1)
    px = rng.integers(0, 256, (1, 1, 3), dtype=np.uint8)
    np.testing.assert_array_equal(N.gaussian_blur(px, 5, 1.0), O.gaussian_blur(px, 5, 1.0))
    np.testing.assert_array_equal(N.color_shift_rgb(px, 100), O.color_shift_rgb(px, 100))
    np.testing.assert_array_equal(N.resize_cubic(px, (7, 5)), O.resize_cubic(px, (7, 5)))
    plane = rng.integers(0, 256, (9, 11), dtype=np.uint8)
    np.testing.assert_array_equal(N.resize_cubic(plane, (1, 1)), O.resize_cubic(plane, (1, 1)))
    np.testing.assert_array_equal(N.fill_poly_mask((1, 1), [(0, 0)]), np.ones((1, 1), np.uint8))
    mx = np.zeros((1, 1), np.float32)
    np.testing.assert_array_equal(N.remap(px, mx, mx), px)


def test_side_limit_is_enforced(N):
    """cv.remap's int16 source coordinates: sources above 32767 px per side are refused, not mis-addressed."""
    src = np.zeros((1, 40000), np.uint8)
    small = np.zeros((2, 2), np.float32)
    with pytest.raises(N.VkxError):
        N.remap(src, small, small)
    with pytest.raises(N.VkxError):
        N.warp_affine(src, np.eye(2, 3), (4, 4))


def test_large_plane_translation_property(N):
    """16000 x 12000 x 3 uint8 (576 MB, byte offsets beyond 2^29 rows x stride): an integer translation through
    warpAffine moves every pixel exactly; checked against numpy slicing, no oracle."""
    h, w = 12000, 16000
    rng = default_rng(2)
    src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    tx, ty = 37, 1021
    M = np.asarray([[1, 0, tx], [0, 1, ty]], np.float64)
    out = N.warp_affine(src, M, (w, h))
    assert out.shape == src.shape
    np.testing.assert_array_equal(out[ty:, tx:], src[:h - ty, :w - tx])
    assert not out[:ty].any() and not out[:, :tx].any()


def test_large_mask_fill_and_paint(N):
    """A 20000 x 20000 mask: polygon paint + composite touch the far corner (row * stride beyond 2^28)."""
    h = w = 20000
    mask = np.zeros((h, w), np.uint8)
    polys = [np.asarray([(w - 50, h - 60), (w - 1, h - 60), (w - 1, h - 1), (w - 50, h - 1)], np.int32),
             np.asarray([(3, 4), (40, 4), (40, 30), (3, 30)], np.int32)]
    N.paint_polys(polys, mask=mask)
    assert mask[h - 60:, w - 50:].all() and mask[4:31, 3:41].all()
    assert int(mask.sum()) == 60 * 50 + 27 * 38
    layer = N.make_layer((h - 10, w - 10, 10, 10), 1, 7)
    layer2 = N.make_layer((0, 0, 2, 2), 1, 9)
    N.fill(mask, [layer, layer2])
    assert (mask[h - 10:, w - 10:] == 7).all() and (mask[:2, :2] == 9).all()
    assert int(mask.sum()) == 60 * 50 - 100 + 27 * 38 + 700 + 36


def test_torch_device_memory_and_stream_interop(N):
    """The *_dev entry points take any device pointer and enqueue on a borrowed stream: tensors and the current stream
    of PyTorch-ROCm (same HIP runtime in the process) work without a copy through the host."""
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.skip('torch sees no GPU')
    rng = default_rng(5)
    img = rng.integers(0, 256, (300, 420, 3), dtype=np.uint8)
    noise = rng.integers(-40, 40, img.shape).astype(np.int16)
    t_src = torch.from_numpy(img).cuda()
    t_noise = torch.from_numpy(noise).cuda()
    t_mid = torch.empty_like(t_src)
    t_dst = torch.empty_like(t_src)
    ctx = N.Context(0)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        lib = N.lib()
        N.check(lib.vkx_color_shift_rgb_dev(ctx.handle, t_src.data_ptr(), 300, 420, 420 * 3, 91, t_mid.data_ptr(), 420 * 3))
        N.check(lib.vkx_add_noise_i16_dev(ctx.handle, t_mid.data_ptr(), 300, 420, 3, 420 * 3, t_noise.data_ptr(), 420 * 3,
                                          t_dst.data_ptr(), 420 * 3))
        doubled = t_dst.to(torch.int32) * 2          # a torch op ordered after the kernels on the same stream
    side.synchronize()
    ctx.set_stream(None)
    want = O.add_noise_i16(O.color_shift_rgb(img, 91), noise)
    np.testing.assert_array_equal(t_dst.cpu().numpy(), want)
    np.testing.assert_array_equal(doubled.cpu().numpy(), want.astype(np.int32) * 2)
    ctx.close()


def test_warp_affine_rgb_on_misaligned_device_planes():
    """The RGB warp reads its taps as aligned dwords from the aligned-down base of the source (csrc/remap.hip): sources that start 1, 2
    and 3 bytes into a device buffer, with a pitch that is not a multiple of 4, ending exactly at the end of the allocation; rotations,
    shears and a shrink, against the oracle."""
    import ctypes
    import oracle as O
    from vkit_amd import _native as N
    ctx = N.default_ctx()
    rng = np.random.default_rng(12)
    for k, (h, w, pad) in enumerate([(61, 83, 0), (40, 129, 5), (97, 64, 2), (33, 47, 7)]):
        pitch = w * 3 + pad
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        host = np.zeros((h, pitch), np.uint8)
        host[:, :w * 3] = img.reshape(h, w * 3)
        for off in (0, 1, 2, 3):
            nbytes = off + (h - 1) * pitch + w * 3           # the last row ends the allocation
            flat = np.zeros(nbytes, np.uint8)
            for y in range(h):
                flat[off + y * pitch: off + y * pitch + w * 3] = host[y, :w * 3]
            d_src = ctx.malloc(nbytes)
            ctx.upload(d_src, flat)
            for M in ([0.9, -0.35, 7.0, 0.3, 1.05, -4.0], [1.0, 0.0, 0.0, 0.0, 1.0, 0.0], [0.5, 0.8, 3.0, -0.8, 0.5, 40.0],
                      [1.7, 0.0, -20.5, 0.2, 1.3, 2.25]):
                dh, dw = h + 11, w + 6
                d_dst = ctx.malloc(dh * dw * 3)
                Mc = (ctypes.c_double * 6)(*M)
                N.check(N.lib().vkx_warp_affine_u8_dev(ctx.handle, ctypes.c_void_p(d_src + off), h, w, 3, pitch, Mc, ctypes.c_void_p(d_dst),
                                                      dh, dw, dw * 3))
                got = np.empty((dh, dw, 3), np.uint8)
                ctx.download(d_dst, got)
                want = O.warp_affine(img, np.asarray(M, np.float64).reshape(2, 3), (dw, dh))
                assert (got == want).all(), (k, off, M)
                ctx.free(d_dst)
            ctx.free(d_src)
