"""Error exits of the three-stream chain call (vkx_chain_rgb_batch_np_dev, csrc/chain.hip) and the stream fallback of an
``add_config`` item (ChainBatch._verify_streams): what the context and the batch look like AFTER something went wrong."""
import os

import numpy as np
import pytest
from numpy.random import default_rng

import oracle as O

pytestmark = pytest.mark.gpu


def _items(n, seed):
    from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam
    rng = default_rng(seed)
    out = []
    for k in range(n):
        shape = (int(rng.integers(80, 260)), int(rng.integers(80, 260)))
        cfg = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), int(rng.integers(1, 11)))(shape, rng)
        out.append((rng.integers(0, 256, shape + (3,), dtype=np.uint8), cfg))
    return out


def _oracle_chain(image, cfg, seed, std=9.0):
    from vkit_amd.mechanism import distortion as D
    st = D.camera_cubic_curve.generate_state(cfg, image.shape[:2])
    mx, my = O.grid_to_map(st.src_image_grid.vertices, st.dst_image_grid.vertices, st.result_shape)
    ref = O.color_shift_rgb(O.gaussian_blur(O.remap(image, mx, my), 5, 1.0), 37)
    plane = np.round(default_rng(seed).normal(0, std, tuple(st.result_shape) + (3,))).astype(np.int16)
    return O.add_noise_i16(ref, plane)


def test_failing_chunk_leaves_the_context_usable():
    """A failure in chunk 1 of the three-stream call (draw(0), draw(1) in flight on the compute stream, post(0), post(1) on aux, the
    cell setup on the side stream): the call reports it, ctx->stream is the compute stream again and ordered after both side
    streams -- a clean call on the SAME context right behind it equals the oracle."""
    from vkit_amd import _native as N
    from vkit_amd.batch import ChainBatch
    ctx = N.Context(N.default_device())
    items = _items(32, 11)           # >= 16 jobs: two chunks; >= 16 images: the setup runs on the side stream

    def batch_of():
        b = ChainBatch(ctx)
        for k, (image, cfg) in enumerate(items):
            b.add_config(image, cfg, blur_sigma=1.0, hue_delta=37, noise_std=9.0, noise_rng=default_rng(300 + k))
        return b

    warm = batch_of()
    warm.run()                       # first run: streams verified, side streams exist
    ctx.sync()
    os.environ['VKX_DEBUG_FAIL_CHUNK'] = '1'
    try:
        with pytest.raises(Exception) as info:
            warm.run()
        assert 'VKX_DEBUG_FAIL_CHUNK' in str(info.value)
        # straight behind the failed call, no host synchronisation in between
        os.environ.pop('VKX_DEBUG_FAIL_CHUNK')
        warm.run()
        got = [warm.result(k) for k in (0, 15, 16, 31)]
    finally:
        os.environ.pop('VKX_DEBUG_FAIL_CHUNK', None)
    for k, g in zip((0, 15, 16, 31), got):
        want = _oracle_chain(items[k][0], items[k][1], 300 + k)
        assert g.shape == want.shape and (g == want).all(), k
    # and a failure in chunk 0, then another batch on the same context
    os.environ['VKX_DEBUG_FAIL_CHUNK'] = '0'
    try:
        with pytest.raises(Exception):
            warm.run()
    finally:
        os.environ.pop('VKX_DEBUG_FAIL_CHUNK', None)
    other = batch_of()
    other.run()
    want = _oracle_chain(items[7][0], items[7][1], 307)
    assert (other.result(7) == want).all()
    warm.close()
    other.close()
    ctx.close()


def test_stream_fallback_of_a_config_item_survives_later_runs():
    """An ``add_config`` item whose noise stream the device declares ambiguous (forced: VKX_NP_DEBUG_WIDE_MARGIN on its job) gets a
    host-drawn int16 plane.  ``_build_states`` rewrites the item's noise pointer on every run: the plane has to stay what the chain
    reads on the re-run and on every later run, and a replaced config (new shape) makes it a stream again."""
    from vkit_amd.batch import ChainBatch
    from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam
    items = _items(5, 23)
    batch = ChainBatch()
    for k, (image, cfg) in enumerate(items):
        batch.add_config(image, cfg, blur_sigma=1.0, hue_delta=37, noise_std=9.0, noise_rng=default_rng(500 + k))
    batch.debug_kind_flags = {2: 0x100}          # VKX_NP_DEBUG_WIDE_MARGIN
    for _ in range(4):
        batch.run()
        assert batch.stream_fallbacks == 1
        for k in (1, 2, 3):
            want = _oracle_chain(items[k][0], items[k][1], 500 + k)
            got = batch.result(k)
            assert got.shape == want.shape and (got == want).all(), k
    # a new config for the item: a new shape, the stream is drawn on the device again (flag off) and equals numpy's
    batch.debug_kind_flags = {}
    image = items[2][0]
    new_cfg = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), 9)(image.shape[:2], default_rng(77))
    batch.set_config(2, new_cfg)
    for _ in range(2):
        batch.run()
        assert (batch.result(2) == _oracle_chain(image, new_cfg, 502)).all()
        assert (batch.result(4) == _oracle_chain(items[4][0], items[4][1], 504)).all()
    assert batch.stream_fallbacks == 1
    batch.close()
