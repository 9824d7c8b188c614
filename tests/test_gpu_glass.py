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
