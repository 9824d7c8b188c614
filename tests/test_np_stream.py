"""The restatement of numpy's Generator streams (oracle/np_random.c) against numpy itself: values AND generator state.
The noise operators of the reference draw from these streams (photometric/noise.py:44-54, 100-157, 160-190)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _words_after(rng):
    return O.np_state_words(rng)


@pytest.mark.parametrize('seed', [0, 1, 7, 12345, 2 ** 63 + 5])
def test_normal_stream_matches_numpy(seed):
    rng = np.random.default_rng(seed)
    words = O.np_state_words(rng)
    n = 300_000
    want = rng.normal(0, 10.0, n)
    got, after, used = O.np_normal(words, n, 0.0, 10.0)
    assert (got.view(np.uint64) == want.view(np.uint64)).all()
    assert (after[:2] == _words_after(rng)[:2]).all()
    assert (O.np_advance(words, int(used))[:2] == after[:2]).all()
    assert n < used < 1.04 * n


def test_normal_tail_and_wedge_are_exercised():
    # 3M samples: ~750 tail samples (|x| > 3.654) and ~45 000 wedge tests
    rng = np.random.default_rng(99)
    words = O.np_state_words(rng)
    n = 3_000_000
    want = rng.normal(0, 1.0, n)
    got, after, _ = O.np_normal(words, n, 0.0, 1.0)
    assert (np.abs(want) > 3.6541528853610088).sum() > 300
    assert (got.view(np.uint64) == want.view(np.uint64)).all()
    assert (after[:2] == _words_after(rng)[:2]).all()


@pytest.mark.parametrize('std', [0.5, 10.0, 25.5, 255.0])
def test_rounded_int16_plane(std):
    rng = np.random.default_rng(int(std * 10))
    words = O.np_state_words(rng)
    shape = (97, 131, 3)
    want = np.round(rng.normal(0, std, shape)).astype(np.int16)
    got, after, _ = O.np_normal_i16(words, want.size, std)
    assert (got.reshape(shape) == want).all()
    assert (after[:2] == _words_after(rng)[:2]).all()


def test_uniform_and_choice():
    rng = np.random.default_rng(5)
    words = O.np_state_words(rng)
    want = rng.random(10_000)
    got, after, used = O.np_random(words, 10_000)
    assert (got == want).all() and used == 10_000
    assert (after[:2] == _words_after(rng)[:2]).all()

    p = [1 - 0.07 - 0.02, 0.07, 0.02]
    rng = np.random.default_rng(6)
    words = O.np_state_words(rng)
    want = rng.choice((0, 1, 2), size=(61, 47), p=p)
    cdf = np.cumsum(np.array(p, np.float64))
    cdf /= cdf[-1]
    got, after, _ = O.np_choice_cdf(words, want.size, cdf)
    assert (got.reshape(want.shape) == want).all()
    assert (after[:2] == _words_after(rng)[:2]).all()


def test_stream_continues_after_other_draws():
    # the operator contract: a generator that has already been used (incl. a buffered 32-bit half) keeps its stream
    rng = np.random.default_rng(3)
    rng.integers(0, 10, 5, dtype=np.int32)
    rng.random(3)
    words = O.np_state_words(rng)
    want = rng.normal(0, 3.0, 1000)
    got, _, _ = O.np_normal(words, 1000, 0.0, 3.0)
    assert (got == want).all()


def test_table_header_matches_installed_numpy():
    archive = os.path.join(os.path.dirname(np.__file__), 'random', 'lib', 'libnpyrandom.a')
    if not os.path.exists(archive):
        pytest.skip('numpy ships no libnpyrandom.a here')
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'np_tables.py'), '--check'], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stdout + proc.stderr
