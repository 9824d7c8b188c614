"""The numpy Generator streams drawn on the device (include/vkx.h: vkx_np_*) against numpy ITSELF: every value and the
generator state after the call.  Reference call sites: photometric/noise.py:44-54, 100-157, 160-190."""
import ctypes

import numpy as np
import pytest

from vkit_amd import _native as N
from vkit_amd.mechanism.distortion import gaussion_noise, impulse_noise, speckle_noise
from vkit_amd.mechanism.distortion.photometric.noise import GaussionNoiseConfig, ImpulseNoiseConfig, SpeckleNoiseConfig
from vkit_amd.element import Image

pytestmark = pytest.mark.gpu


def _same_state(a, b):
    return a.bit_generator.state == b.bit_generator.state


@pytest.mark.parametrize('seed,n,std', [(0, 1, 1.0), (1, 63, 10.0), (2, 64, 10.0), (3, 1023, 2.5), (4, 1024, 2.5), (5, 1025, 40.0),
                                        (6, 70_001, 10.0), (7, 3 * 1024 * 1024, 0.0), (8, 2047 * 2049 * 3, 10.0),
                                        (9, 12_582_912, 255.0)])
def test_rounded_normal_plane_matches_numpy(seed, n, std):
    rng, ref = np.random.default_rng(seed), np.random.default_rng(seed)
    want = np.round(ref.normal(0, std, n)).astype(np.int16)
    got = N.np_normal_i16((n,), std, rng)
    assert got is not None
    assert (got == want).all()
    assert _same_state(rng, ref)


@pytest.mark.parametrize('seed,n,std', [(20, 1, 1.0), (21, 2, 3.0), (22, 3005, 10.0), (23, 3007, 10.0), (24, 70_001, 10.0),
                                        (25, 3 * 1024 * 1024, 0.0), (26, 2047 * 2049 * 3, 12.0), (27, 12_582_912, 255.0)])
def test_tile_buffer_stands_for_the_plane(seed, n, std):
    """VKX_NP_NORMAL_TILES: the samples stay in the generator's tile slots; the table places them.  The buffer read back through
    the documented layout, and through vkx_np_tiles_expand_dev, is numpy's plane; every slot ends with the first two samples of
    its successor; the draw count moves the generator to numpy's state."""
    ctx = N.default_ctx()
    rng, ref = np.random.default_rng(seed), np.random.default_rng(seed)
    want = np.round(ref.normal(0, std, n)).astype(np.int16)
    tiles, slot, table_off, slots_off, nbytes = N.np_tiles_layout(n)
    buf = np.zeros(nbytes, np.uint8)
    job = N.np_job(N.NP_NORMAL_TILES, N.np_stream(rng), n, std, dst=N._ptr(buf))
    res = N.VkxNpResult()
    N.check(N.lib().vkx_np_draw(ctx.handle, ctypes.byref(job), ctypes.byref(res)))
    assert res.flags == 0 and res.samples >= n
    assert (N.np_tiles_plane(buf, n) == want).all()
    N.np_consume(rng, res.draws)
    assert _same_state(rng, ref)
    header = buf[:16].view(np.uint32)
    assert header[0] == tiles and header[1] == slot and int(buf[8:16].view(np.uint64)[0]) == res.samples
    table = buf[table_off:table_off + 8 * (tiles + 1)].view(np.uint32).reshape(tiles + 1, 2)
    slots = buf[slots_off:slots_off + tiles * slot * 2].view(np.int16).reshape(tiles, slot)
    assert table[0, 0] == 0 and table[tiles, 0] == res.samples and (np.diff(table[:, 0].astype(np.int64)) >= 0).all()
    for t in range(tiles - 1):
        nxt = int(table[t + 1, 0])
        if nxt + 2 > n:
            break
        end = int(table[t, 1]) + nxt - int(table[t, 0])
        assert (slots[t, end:end + 2] == want[nxt:nxt + 2]).all(), t
    # on the device
    dbuf = ctx.to_device(buf)
    plane = ctx.dev_empty((n,), np.int16)
    N.check(N.lib().vkx_np_tiles_expand_dev(ctx.handle, ctypes.c_void_p(dbuf.ptr), n, ctypes.c_void_p(plane.ptr)))
    assert (plane.host() == want).all()


def test_tile_buffers_of_a_batch():
    """Several VKX_NP_NORMAL_TILES streams of one device call, sizes around the tile and chunk boundaries."""
    ctx = N.default_ctx()
    sizes = [1, 5, 3006, 3072, 6013, 50_000, 123_457] + [40_000 + 977 * k for k in range(70)]
    B = len(sizes)
    jobs = (N.VkxNpJob * B)()
    res = N.NpResults(ctx, B)
    bufs = []
    for i, n in enumerate(sizes):
        d = ctx.dev_empty((N.np_tiles_layout(n)[4],), np.uint8)
        bufs.append(d)
        jobs[i] = N.np_job(N.NP_NORMAL_TILES, N.np_stream(np.random.default_rng(700 + i)), n, 3.0 + i % 7, dst=d.ptr)
    N.check(N.lib().vkx_np_draw_batch_dev(ctx.handle, jobs, B, res.array))
    ctx.sync()
    for i, n in enumerate(sizes):
        ref = np.random.default_rng(700 + i)
        want = np.round(ref.normal(0, 3.0 + i % 7, n)).astype(np.int16)
        assert res[i].flags == 0
        assert (N.np_tiles_plane(bufs[i].host(), n) == want).all(), (i, n)
        chk = np.random.default_rng(700 + i)
        N.np_consume(chk, res[i].draws)
        assert _same_state(chk, ref)


def test_stream_continues_across_calls_and_other_draws():
    rng, ref = np.random.default_rng(77), np.random.default_rng(77)
    for r in (rng, ref):
        r.integers(0, 10, 3, dtype=np.int32)       # leaves a buffered 32-bit half behind
        r.random(5)
    for n, std in ((1000, 3.0), (70_000, 10.0), (5, 1.0), (300_000, 25.0)):
        want = np.round(ref.normal(0, std, n)).astype(np.int16)
        got = N.np_normal_i16((n,), std, rng)
        assert (got == want).all() and _same_state(rng, ref)
    # the buffered half is still there: the next 32-bit draw agrees
    assert (rng.integers(0, 1 << 30, 4, dtype=np.int32) == ref.integers(0, 1 << 30, 4, dtype=np.int32)).all()


def test_operators_draw_on_the_device_and_match_the_host_formulas():
    img = np.random.default_rng(5).integers(0, 256, (301, 211, 3), dtype=np.uint8)
    rng, ref = np.random.default_rng(11), np.random.default_rng(11)

    got = N.np_gaussion_noise(img, 12.5, rng)
    want = np.clip(img.astype(np.int16) + np.round(ref.normal(0, 12.5, img.shape)).astype(np.int16), 0, 255).astype(np.uint8)
    assert (got == want).all() and _same_state(rng, ref)

    got = N.np_speckle_noise(img, 0.3, rng)
    m = img.astype(np.float32)
    want = np.clip(m + m * ref.normal(0, 0.3, m.shape), 0, 255).astype(np.uint8)
    assert (got == want).all() and _same_state(rng, ref)

    for sel_img in (img, img[:, :, 0].copy()):
        got = N.np_impulse_noise(sel_img, 0.05, 0.03, rng)
        mask = ref.choice((0, 1, 2), size=sel_img.shape[:2], p=[1 - 0.05 - 0.03, 0.05, 0.03])
        want = sel_img.copy()
        want[mask == 1] = 255
        want[mask == 2] = 0
        assert (got == want).all() and _same_state(rng, ref)


def test_speckle_with_black_pixels_stays_on_the_device():
    """A zero pixel is zero whatever the noise: a tail draw landing on one is not an ambiguity (several seeds, 3 M samples each:
    ~800 tail draws per plane, a third of them on black pixels here)."""
    img = np.random.default_rng(2).integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
    img[::3] = 0
    for seed in range(4):
        rng, ref = np.random.default_rng(seed), np.random.default_rng(seed)
        got = N.np_speckle_noise(img, 0.25, rng)
        assert got is not None, seed
        m = img.astype(np.float32)
        want = np.clip(m + m * ref.normal(0, 0.25, m.shape), 0, 255).astype(np.uint8)
        assert (got == want).all() and _same_state(rng, ref)


def test_distortion_operators_take_the_device_path(monkeypatch):
    """gaussion / speckle / impulse through Distortion.distort: same pixels and same generator state as the host formulas,
    and the device path is the one that ran."""
    img = Image(mat=np.random.default_rng(1).integers(0, 256, (257, 199, 3), dtype=np.uint8))
    taken = []
    inner = N.np_draw

    def counted(*args, **kwargs):
        out = inner(*args, **kwargs)
        taken.append(out is not None)
        return out

    monkeypatch.setattr(N, 'np_draw', counted)
    for op, config, formula in (
        (gaussion_noise, GaussionNoiseConfig(std=9.0),
         lambda r, m: np.clip(m.astype(np.int16) + np.round(r.normal(0, 9.0, m.shape)).astype(np.int16), 0, 255).astype(np.uint8)),
        (speckle_noise, SpeckleNoiseConfig(std=0.2),
         lambda r, m: np.clip(m.astype(np.float32) + m.astype(np.float32) * r.normal(0, 0.2, m.shape), 0, 255).astype(np.uint8)),
    ):
        # the operator contract (distortion/interface.py:261-347): the pixels come from a private generator holding the
        # caller's state, the caller's own stream moves on by one rng.random()
        rng, ref = np.random.default_rng(3), np.random.default_rng(3)
        out = op.distort(config, image=img, rng=rng)
        want = formula(np.random.default_rng(3), img.mat)
        assert (out.image.mat == want).all()
        ref.random()
        assert _same_state(rng, ref)
        # replay from the recorded state gives the same pixels
        again = op.distort(config, image=img)
        assert (again.image.mat == want).all()

    rng = np.random.default_rng(4)
    config = ImpulseNoiseConfig(prob_salt=0.04, prob_pepper=0.06)
    out = impulse_noise.distort(config, image=img, rng=rng)
    mask = np.random.default_rng(4).choice((0, 1, 2), size=img.shape, p=[1 - 0.04 - 0.06, 0.04, 0.06])
    want = img.mat.copy()
    want[mask == 1] = 255
    want[mask == 2] = 0
    assert (out.image.mat == want).all()
    assert taken == [True] * 5


def test_other_bit_generators_and_forced_host_mode_fall_back(monkeypatch):
    img = np.random.default_rng(5).integers(0, 256, (64, 64, 3), dtype=np.uint8)
    rng = np.random.Generator(np.random.Philox(1))
    assert N.np_gaussion_noise(img, 5.0, rng) is None
    monkeypatch.setenv('VKX_HOST_RNG', '1')
    assert N.np_gaussion_noise(img, 5.0, np.random.default_rng(0)) is None
    # the operator still works through the host draw
    out = gaussion_noise.distort(GaussionNoiseConfig(std=5.0), image=Image(mat=img), rng=np.random.default_rng(8))
    want = np.clip(img.astype(np.int16) + np.round(np.random.default_rng(8).normal(0, 5.0, img.shape)).astype(np.int16), 0, 255)
    assert (out.image.mat == want.astype(np.uint8)).all()


def test_ambiguity_flag_is_raised_and_leaves_the_generator_alone():
    """VKX_NP_DEBUG_WIDE_MARGIN declares every wedge test ambiguous: the job reports it, np_draw then refuses the result."""
    ctx = N.default_ctx()
    rng = np.random.default_rng(9)
    before = rng.bit_generator.state
    n = 100_000
    dst = ctx.pinned_empty((n,), np.int16)
    job = N.np_job(N.NP_NORMAL_I16 | 0x100, N.np_stream(rng), n, 10.0, dst=N._ptr(dst))
    res = N.VkxNpResult()
    N.check(N.lib().vkx_np_draw(ctx.handle, ctypes.byref(job), ctypes.byref(res)))
    assert res.flags & N.NP_AMBIGUOUS
    assert rng.bit_generator.state == before
    # the values themselves are still numpy's (the flag is conservative)
    assert (dst == np.round(np.random.default_rng(9).normal(0, 10.0, n)).astype(np.int16)).all()


def test_batch_of_streams_device_resident():
    ctx = N.default_ctx()
    B, n = 7, 1_000_003
    jobs = (N.VkxNpJob * B)()
    res = (N.VkxNpResult * B)()
    bufs = []
    for i in range(B):
        p = ctx.malloc(n * 2)
        bufs.append(p)
        jobs[i] = N.np_job(N.NP_NORMAL_I16, N.np_stream(np.random.default_rng(100 + i)), n - i, 4.0 + i, dst=p)
    N.check(N.lib().vkx_np_draw_batch_dev(ctx.handle, jobs, B, res))
    ctx.sync()
    for i in range(B):
        ref = np.random.default_rng(100 + i)
        want = np.round(ref.normal(0, 4.0 + i, n - i)).astype(np.int16)
        got = np.empty(n - i, np.int16)
        ctx.download(bufs[i], got)
        assert res[i].flags == 0
        assert (got == want).all()
        st = ref.bit_generator.state['state']
        assert N.pcg64_jump(*N.np_stream(np.random.default_rng(100 + i)), res[i].draws) == st['state']
        ctx.free(bufs[i])


def test_large_call_runs_in_pipelined_chunks():
    """A call of 64 jobs or more runs as chunks of 32 on two streams with alternating scratch slots (nprand.hip): ragged
    streams, int16 planes and the in-place add onto uint8 pixels, results through a page-locked buffer (written by the last
    workgroup of each chunk) and through a pageable one (copied back), twice in a row on the same context."""
    ctx = N.default_ctx()
    rng = np.random.default_rng(77)
    B = 83
    sizes = [int(v) for v in rng.integers(1, 400_000, B)]
    sizes[5], sizes[40], sizes[82] = 1, 1024, 1_500_001
    stds = [float(v) for v in rng.uniform(0.5, 30.0, B)]
    for kind in (N.NP_NORMAL_I16, N.NP_NORMAL_ADD_U8):
        for pinned in (True, False):
            jobs = (N.VkxNpJob * B)()
            res = N.NpResults(ctx, B) if pinned else None
            res_array = res.array if pinned else (N.VkxNpResult * B)()
            planes, pixels = [], []
            for i in range(B):
                if kind == N.NP_NORMAL_I16:
                    d = ctx.dev_empty((sizes[i],), np.int16)
                    jobs[i] = N.np_job(kind, N.np_stream(np.random.default_rng(300 + i)), sizes[i], stds[i], dst=d.ptr)
                else:
                    px = np.random.default_rng(900 + i).integers(0, 256, sizes[i], dtype=np.uint8)
                    d = ctx.to_device(px)
                    pixels.append(px)
                    jobs[i] = N.np_job(kind, N.np_stream(np.random.default_rng(300 + i)), sizes[i], stds[i], src=d.ptr, dst=d.ptr)
                planes.append(d)
            for _rep in range(2 if kind == N.NP_NORMAL_I16 else 1):
                N.check(N.lib().vkx_np_draw_batch_dev(ctx.handle, jobs, B, res_array))
            ctx.sync()
            for i in range(B):
                ref = np.random.default_rng(300 + i)
                noise = np.round(ref.normal(0, stds[i], sizes[i])).astype(np.int16)
                got = planes[i].host()
                if kind == N.NP_NORMAL_I16:
                    assert (got == noise).all(), (i, sizes[i])
                else:
                    assert (got == np.clip(pixels[i].astype(np.int16) + noise, 0, 255).astype(np.uint8)).all(), (i, sizes[i])
                assert res_array[i].flags == 0 and res_array[i].samples >= sizes[i]
                assert N.pcg64_jump(*N.np_stream(np.random.default_rng(300 + i)), res_array[i].draws) == ref.bit_generator.state['state']['state']


def test_a_billion_samples():
    """>= 1e9 samples over seeds / lengths / deviations, every one compared with numpy (about a minute of host draws)."""
    ctx = N.default_ctx()
    total = 0
    chunk = 0
    while total < 1_000_000_000:
        n = 83_886_080 + 1021 * chunk        # ~84 M samples per stream, varying length
        std = (0.7, 3.0, 10.0, 25.0, 60.0, 254.0)[chunk % 6]
        rng, ref = np.random.default_rng(9000 + chunk), np.random.default_rng(9000 + chunk)
        got = N.np_normal_i16((n,), std, rng, ctx)
        want = np.round(ref.normal(0, std, n)).astype(np.int16)
        assert got is not None, chunk
        assert (got == want).all(), chunk
        assert _same_state(rng, ref), chunk
        total += n
        chunk += 1
        del got, want
