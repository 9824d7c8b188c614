"""Development probe of the device numpy streams: correctness against numpy and kernel timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vkit_amd import _native as N

ctx = N.default_ctx()
ok_all = True
QUICK = len(sys.argv) > 2
for seed, n, std in [] if QUICK else [(0, 1000, 10.0), (1, 70000, 10.0), (2, 1 << 20, 3.0), (3, 12_582_912, 10.0), (4, 5, 1.0), (5, 2048 * 2048 * 3 + 17, 25.0)]:
    rng = np.random.default_rng(seed)
    ref = np.random.default_rng(seed)
    want = np.round(ref.normal(0, std, n)).astype(np.int16)
    got = N.np_normal_i16((n,), std, rng, ctx)
    if got is None:
        print(seed, n, 'device path declined'); ok_all = False; continue
    same = (got == want).all()
    st = rng.bit_generator.state == ref.bit_generator.state
    print(seed, n, std, 'values', same, 'state', st, flush=True)
    if not same:
        bad = np.flatnonzero(got != want)
        print('  first diffs', bad[:10], got[bad[:5]], want[bad[:5]], 'count', bad.size)
    ok_all &= bool(same and st)

# operators
if not QUICK:
    rng = np.random.default_rng(11); ref = np.random.default_rng(11)
    img = np.random.default_rng(5).integers(0, 256, (300, 211, 3), dtype=np.uint8)
    got = N.np_gaussion_noise(img, 12.5, rng, ctx)
    want = np.clip(img.astype(np.int16) + np.round(ref.normal(0, 12.5, img.shape)).astype(np.int16), 0, 255).astype(np.uint8)
    print('gaussion', (got == want).all(), rng.bit_generator.state == ref.bit_generator.state)
    got = N.np_speckle_noise(img, 0.3, rng, ctx)
    m = img.astype(np.float32)
    want = np.clip(m + m * ref.normal(0, 0.3, m.shape), 0, 255).astype(np.uint8)
    print('speckle', (got == want).all(), rng.bit_generator.state == ref.bit_generator.state)
    got = N.np_impulse_noise(img, 0.05, 0.03, rng, ctx)
    mask = ref.choice((0, 1, 2), size=img.shape[:2], p=[1 - 0.05 - 0.03, 0.05, 0.03])
    want = img.copy(); want[mask == 1] = 255; want[mask == 2] = 0
    print('impulse', (got == want).all(), rng.bit_generator.state == ref.bit_generator.state)

# timing: 64 planes of 2048^2 x 3
import ctypes
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = 2147 * 2115 * 3
jobs = (N.VkxNpJob * B)()
res = (N.VkxNpResult * B)()
bufs = []
for i in range(B):
    r = np.random.default_rng(5000 + i)
    if os.environ.get('NP_KIND') == 'tiles':
        p = ctx.malloc(N.np_tiles_layout(n)[4]); bufs.append(p)
        jobs[i] = N.np_job(N.NP_NORMAL_TILES, N.np_stream(r), n, 10.0, dst=p)
        continue
    p = ctx.malloc(n * 2); bufs.append(p)
    jobs[i] = N.np_job(N.NP_NORMAL_I16, N.np_stream(r), n, 10.0, dst=p)
for it in range(2):
    N.check(N.lib().vkx_np_draw_batch_dev(ctx.handle, jobs, B, res))
ctx.sync()
ctx.set_timing(True)
ctx.reset_timings()
for it in range(3):
    N.check(N.lib().vkx_np_draw_batch_dev(ctx.handle, jobs, B, res))
ctx.sync()
for k, (ms, cnt) in ctx.timings().items():
    print(f'{k}: {ms / cnt:.3f} ms x {cnt}')
print('flags', [res[i].flags for i in range(min(B, 8))], 'draws/n', res[0].draws / n)
w = np.round(np.random.default_rng(5003).normal(0, 10.0, n)).astype(np.int16)
if os.environ.get('NP_KIND') == 'tiles':
    raw = np.empty(N.np_tiles_layout(n)[4], np.uint8); ctx.download(bufs[3], raw)
    out = N.np_tiles_plane(raw, n)
else:
    out = np.empty(n, np.int16); ctx.download(bufs[3], out)
print('batch plane 3 exact', (out == w).all())
print('ALL OK' if ok_all else 'FAILURES')
