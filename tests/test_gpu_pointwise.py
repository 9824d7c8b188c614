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
