"""The fog density plane drawn on the device (vkx_fog_field_f32_dev + vkx_fog_stretch_f32_dev, vkit_amd/csrc/fog.hip) against the host
restatement of the reference's generate_diamond_square_mask + stretch (photometric/effect.py), which the reference's golden outputs pin
(tests/test_gpu_pointwise.py): the mask bit for bit and the generator left at the same position, for square and oblong shapes, sizes
on both sides of a power of two, every roughness regime."""
import numpy as np
import pytest
from numpy.random import default_rng

pytestmark = pytest.mark.gpu


def _host_mask(shape, roughness, lo, hi, rng):
    from vkit_amd.mechanism.distortion.photometric.effect import generate_diamond_square_mask
    mask = np.array(generate_diamond_square_mask(shape, roughness, rng), dtype=np.float32)
    mask -= mask.min()
    mask /= mask.max()
    mask *= (hi - lo)
    mask += lo
    return mask


@pytest.mark.parametrize('shape', [(2, 2), (3, 5), (16, 16), (17, 9), (100, 257), (512, 512), (513, 300), (1024, 1024), (700, 1300)])
@pytest.mark.parametrize('roughness', [0.0, 0.35, 0.8, 1.0])
def test_mask_and_stream_position(shape, roughness):
    from vkit_amd import _native as N
    seed = shape[0] * 7 + shape[1] + int(roughness * 100)
    r_np, r_dev = default_rng(seed), default_rng(seed)
    r_np.random(5); r_dev.random(5)
    want = _host_mask(shape, roughness, 0.1, 0.9, r_np)
    got = N.np_fog_mask(shape, roughness, 0.1, 0.9, r_dev)
    assert got is not None
    got = np.asarray(N.host_array(got))
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    assert r_np.bit_generator.state == r_dev.bit_generator.state


def test_the_member_and_other_generators():
    from numpy.random import Generator, Philox
    from vkit_amd import _native as N
    from vkit_amd.element import Image
    from vkit_amd.mechanism import distortion as D
    assert N.np_fog_mask((64, 64), 0.5, 0.0, 1.0, Generator(Philox(1))) is None
    img = default_rng(2).integers(0, 256, (300, 420, 3), dtype=np.uint8)
    cfg = D.FogConfig(roughness=0.6, ratio_max=0.8, ratio_min=0.1)
    a = D.fog.distort(cfg, image=Image(mat=img), rng=default_rng(4)).image.mat
    b = D.fog.distort(cfg, image=Image(mat=img), rng=Generator(Philox(1))).image.mat      # numpy builds the field
    assert a.shape == b.shape == img.shape
    import os
    os.environ['VKX_HOST_RNG'] = '1'
    try:
        c = D.fog.distort(cfg, image=Image(mat=img), rng=default_rng(4)).image.mat        # the same stream, field by numpy
    finally:
        del os.environ['VKX_HOST_RNG']
    np.testing.assert_array_equal(np.asarray(a), np.asarray(c))
