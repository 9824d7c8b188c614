"""glass_blur's shuffle planes built on the device (vkx_glass_round_dev, vkit_amd/csrc/fog.hip) against the host restatement of the
reference's statements (photometric/blur.py: glass_shuffle_planes -- numpy fancy-index tuple assignments with their order of duplicates):
both planes and the generator's position, for several rounds (from the second round on targets collide and chains of swaps form)."""
import numpy as np
import pytest
from numpy.random import default_rng

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape', [(1, 1), (3, 7), (40, 33), (257, 300), (512, 512), (700, 1024)])
@pytest.mark.parametrize('delta,loop', [(1, 1), (1, 4), (2, 3), (4, 6), (7, 2)])
def test_planes_and_stream(shape, delta, loop):
    from vkit_amd import _native as N
    from vkit_amd.mechanism.distortion.photometric.blur import glass_shuffle_planes
    seed = shape[0] * 31 + shape[1] + delta * 7 + loop
    r_np, r_dev = default_rng(seed), default_rng(seed)
    want_y, want_x = glass_shuffle_planes(shape, delta, loop, r_np)
    got_y, got_x = N.glass_shuffle_planes_dev(shape, delta, loop, r_dev)
    np.testing.assert_array_equal(np.asarray(N.host_array(got_y)), want_y)
    np.testing.assert_array_equal(np.asarray(N.host_array(got_x)), want_x)
    assert r_np.bit_generator.state == r_dev.bit_generator.state


def test_member_matches_host_planes(monkeypatch):
    from vkit_amd.element import Image
    from vkit_amd.mechanism import distortion as D
    img = default_rng(2).integers(0, 256, (310, 420, 3), dtype=np.uint8)
    cfg = D.GlassBlurConfig(sigma=1.2, delta=2, loop=4)
    a = np.asarray(D.glass_blur.distort(cfg, image=Image(mat=img), rng=default_rng(4)).image.mat)
    monkeypatch.setenv('VKX_HOST_SHUFFLE', '1')
    b = np.asarray(D.glass_blur.distort(cfg, image=Image(mat=img), rng=default_rng(4)).image.mat)
    np.testing.assert_array_equal(a, b)


def test_argument_errors_are_reported():
    """The new entry points refuse what they cannot take, with a message, instead of launching."""
    import ctypes
    from vkit_amd import _native as N
    ctx = N.Context(0)
    lib = N.lib()
    pos = ctx.dev_empty((8, 8), np.int32)
    jumps = np.zeros(4, np.int32)
    p = ctypes.c_void_p(pos.ptr)
    # a round before vkx_glass_init_dev sized the winner plane of this context
    assert lib.vkx_glass_round_dev(ctx.handle, p, p, 8, 8, 0, 0, 3, 2, 2, N._ptr(jumps), N._ptr(jumps)) == N.ERR_INVALID
    assert b'vkx_glass_init_dev' in lib.vkx_last_error()
    assert lib.vkx_glass_init_dev(ctx.handle, p, p, 8, 8) == 0
    assert lib.vkx_glass_round_dev(ctx.handle, p, p, 8, 8, 0, 0, 3, 4, 4, N._ptr(jumps), N._ptr(jumps)) == N.ERR_INVALID     # lattice outside the plane
    field = ctx.dev_empty((3, 3), np.float32)
    st = (ctypes.c_uint64 * 2)(1, 2)
    nw = np.ones(1)
    corners = np.zeros(4, np.float32)
    consumed = ctypes.c_longlong()
    assert lib.vkx_fog_field_f32_dev(ctx.handle, st, st, 0, N._ptr(nw), N._ptr(corners), ctypes.c_void_p(field.ptr), ctypes.byref(consumed)) == N.ERR_INVALID
    assert lib.vkx_fog_stretch_f32_dev(ctx.handle, ctypes.c_void_p(field.ptr), 3, 2, 2, 2, 2, 1.0, 0.0, ctypes.c_void_p(field.ptr)) == N.ERR_INVALID
    img16 = np.zeros((4, 4), np.int16)
    assert N.np_poisson_u8(img16, default_rng(1)) is None
    ctx.sync()
