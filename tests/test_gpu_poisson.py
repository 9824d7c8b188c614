"""rng.poisson drawn on the device (vkx_np_poisson_u8, vkit_amd/csrc/poisson.hip) against numpy itself -- the reference's
poisson_noise (photometric/noise.py:81-90) is exactly `rng.poisson(mat.astype(float32))` + a clip: values and the position the
generator is left at, for every regime of random_poisson (lam = 0, the multiplication method below 10, PTRS from 10 up), ragged
lengths around the block size, planes and interleaved images, host arrays and device-resident ones."""
import numpy as np
import pytest
from numpy.random import default_rng

pytestmark = pytest.mark.gpu


def _check(img, seed=11, skip=3):
    from vkit_amd import _native as N
    r_np, r_dev = default_rng(seed), default_rng(seed)
    r_np.random(skip); r_dev.random(skip)          # a stream that is not at its origin
    want = np.clip(r_np.poisson(img.astype(np.float32)), 0, 255).astype(np.uint8)
    got = N.np_poisson_u8(img, r_dev)
    assert got is not None, f'device path declined: flags {N.np_poisson_last_flags()}'
    np.testing.assert_array_equal(np.asarray(N.host_array(got)), want)
    assert r_np.bit_generator.state == r_dev.bit_generator.state
    assert r_np.random() == r_dev.random()


@pytest.mark.parametrize('n', [1, 2, 31, 32, 33, 63, 64, 65, 1000, 8191, 8192, 8193, 100003])
def test_ragged_lengths(n):
    _check(default_rng(n).integers(0, 256, (n,), dtype=np.uint8), seed=n)


@pytest.mark.parametrize('lo,hi', [(0, 1), (0, 2), (1, 10), (9, 11), (10, 11), (10, 30), (0, 13), (200, 256), (255, 256), (0, 256)])
def test_every_regime(lo, hi):
    _check(default_rng(lo * 256 + hi).integers(lo, hi, (301, 257, 3), dtype=np.uint8), seed=hi)


def test_page_like_image_and_plane():
    g = default_rng(3)
    page = np.full((768, 1024, 3), 255, np.uint8)
    for k in range(0, 700, 40):
        page[k + 8:k + 24, 32:-32] = g.integers(0, 60, (16, 960, 3), dtype=np.uint8) * (g.random((16, 960, 1)) < 0.4)
    _check(page)
    _check(g.integers(0, 256, (517, 1031), dtype=np.uint8))
    _check(np.ascontiguousarray(page[:, :, 0]))


def test_long_runs_of_one_value_cross_superblocks():
    # 8192 elements per superblock at most: constant stretches, a step in the middle of one, zeros that take no draws at all
    img = np.concatenate([np.full(20000, 255, np.uint8), np.zeros(9000, np.uint8), np.full(30000, 9, np.uint8), np.full(12345, 10, np.uint8),
                          np.full(7, 3, np.uint8)])
    _check(img)


def test_many_seeds_small_images():
    for seed in range(40):
        _check(default_rng(1000 + seed).integers(0, 256, (64, 48, 3), dtype=np.uint8), seed=seed, skip=seed)


def test_resident_input_and_the_distortion_member():
    from vkit_amd import _native as N
    from vkit_amd.element import Image
    from vkit_amd.mechanism import distortion as D
    img = default_rng(5).integers(0, 256, (400, 300, 3), dtype=np.uint8)
    r_np = default_rng(9)
    want = np.clip(r_np.poisson(img.astype(np.float32)), 0, 255).astype(np.uint8)
    out = D.poisson_noise.distort({}, image=Image(mat=img), rng=default_rng(9)).image
    np.testing.assert_array_equal(out.mat, want)
    with N.resident(True):
        r2 = default_rng(9)
        got = N.np_poisson_u8(N.default_ctx().to_device(img), r2)
        assert isinstance(got, N.DevArray)
        np.testing.assert_array_equal(np.asarray(N.host_array(got)), want)
    assert r_np.bit_generator.state == r2.bit_generator.state


def test_other_bit_generators_take_numpy(monkeypatch):
    from vkit_amd import _native as N
    from numpy.random import Generator, Philox
    img = default_rng(5).integers(0, 256, (50, 60, 3), dtype=np.uint8)
    assert N.np_poisson_u8(img, Generator(Philox(3))) is None
    monkeypatch.setenv('VKX_HOST_RNG', '1')
    assert N.np_poisson_u8(img, default_rng(3)) is None


def test_a_refused_image_leaves_the_generator_alone_and_numpy_draws():
    """VKX_PZ_SIGMAS=0.3 makes the windows far too narrow (a separate process: the library reads it once): the device must refuse
    (WINDOW flag), leave the generator where it was, and poisson_noise must come out as numpy's all the same."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np
from numpy.random import default_rng
from vkit_amd import _native as N
from vkit_amd.element import Image
from vkit_amd.mechanism import distortion as D
img = default_rng(5).integers(0, 256, (300, 400, 3), dtype=np.uint8)
r = default_rng(9)
before = r.bit_generator.state
assert N.np_poisson_u8(img, r) is None and N.np_poisson_last_flags() & 4, N.np_poisson_last_flags()
assert r.bit_generator.state == before
want = np.clip(default_rng(9).poisson(img.astype(np.float32)), 0, 255).astype(np.uint8)
out = D.poisson_noise.distort({}, image=Image(mat=img), rng=default_rng(9)).image
assert np.array_equal(out.mat, want)
print('refused and redrawn')
'''
    env = dict(os.environ, VKX_PZ_SIGMAS='0.3')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'refused and redrawn' in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize('depth', ['1', '3'])
def test_other_pipeline_depths(depth):
    """VKX_PZ_DEPTH (read once per process): launches strictly in sequence, and three superblock launches in flight."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np
from numpy.random import default_rng
from vkit_amd import _native as N
g = default_rng(1)
for img in (g.integers(0, 256, (300, 400, 3), dtype=np.uint8), np.full((257, 513, 3), 250, np.uint8), g.integers(0, 14, (200, 300, 3), dtype=np.uint8)):
    r_np, r_dev = default_rng(9), default_rng(9)
    want = np.clip(r_np.poisson(img.astype(np.float32)), 0, 255).astype(np.uint8)
    got = N.np_poisson_u8(img, r_dev)
    assert got is not None and np.array_equal(np.asarray(N.host_array(got)), want) and r_np.bit_generator.state == r_dev.bit_generator.state
print('depth ok')
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, VKX_PZ_DEPTH=depth), cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'depth ok' in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize('env', [{'VKX_PZ_GRID': '1'}, {'VKX_PZ_GRID': '3', 'VKX_PZ_DEPTH': '3'}, {'VKX_PZ_GRID': '7', 'VKX_PZ_DEPTH': '1'},
                                 {'VKX_PZ_G_GLOBAL': '1'}, {'VKX_PZ_G_GLOBAL': '1', 'VKX_PZ_DEPTH': '3', 'VKX_PZ_GRID': '5'}])
def test_persistent_kernel_structure(env):
    """The one-launch kernel (round 5): a workgroup waits only for results of lower tickets, so ANY number of persistent workgroups must
    finish -- one workgroup does every block, group walk and chain itself, a handful interleave superblocks at every depth --; the top
    chain reading its G rows from global memory (VKX_PZ_G_GLOBAL: the path taken when they outgrow LDS); more superblocks than ring
    slots; groups of fewer than sixteen blocks; dark images whose superblocks close early."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np
from numpy.random import default_rng
from vkit_amd import _native as N
g = default_rng(2)
page = np.full((400, 520, 3), 255, np.uint8)
page[40:60, 30:480] = g.integers(0, 80, (20, 450, 3), dtype=np.uint8)
cases = (g.integers(0, 256, (260, 333, 3), dtype=np.uint8), page, g.integers(0, 14, (150, 210, 3), dtype=np.uint8),
         np.full((97,), 10, np.uint8), np.full((32 * 16 + 5,), 200, np.uint8), np.zeros((700,), np.uint8), np.full((32 * 17,), 9, np.uint8))
for k, img in enumerate(cases):
    r_np, r_dev = default_rng(30 + k), default_rng(30 + k)
    want = np.clip(r_np.poisson(img.astype(np.float32)), 0, 255).astype(np.uint8)
    got = N.np_poisson_u8(img, r_dev)
    assert got is not None, (k, N.np_poisson_last_flags())
    assert np.array_equal(np.asarray(N.host_array(got)), want) and r_np.bit_generator.state == r_dev.bit_generator.state, k
print('structure ok')
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, **env), cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'structure ok' in out.stdout, out.stdout + out.stderr
