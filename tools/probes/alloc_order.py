"""Does the speed of the C3 step depend on WHEN its buffers were allocated?  (bench.py's lattices_resident leg ran 12 % slower
than the same step in the main batch.)  Batches of 96 pages, each timed alone; a second batch allocated while the first is alive; the
first again; a third on another context."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from vkit_amd import _native as N
from vkit_amd.batch import ChainBatch

B, size = 96, 2048
states = [bench.make_state(j, size) for j in range(B)]
images = [np.random.default_rng(1000 + j).integers(0, 256, (size, size, 3), dtype=np.uint8) for j in range(8)]


def build(ctx):
    b = ChainBatch(ctx)
    for j in range(B):
        b.add(images[j % 8], states[j], blur_sigma=1.0, hue_delta=37, noise_std=10.0, noise_rng=np.random.default_rng(5000 + j))
    return b


def timed(b, label, steps=12):
    b.run(); b.ctx.sync()
    b.ctx.set_timing(2); b.ctx.reset_timings()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
    b.ctx.sync()
    dt = (time.perf_counter() - t0) / steps * 1e3
    k = {n: round(v[0] / steps, 3) for n, v in b.ctx.timings().items()}
    b.ctx.set_timing(False)
    print(f'{label:40s} {dt:7.3f} ms/step  {k}', flush=True)


c1, c2 = N.Context(0), N.Context(0)
a = build(c1); timed(a, 'A (first allocation, ctx 1)')
b = build(c1); timed(b, 'B (second, same ctx, A alive)')
timed(a, 'A again')
c = build(c2); timed(c, 'C (third, ctx 2)')
a.close(); b.close()
timed(c, 'C after A, B freed')
d = build(c1); timed(d, 'D (ctx 1 after frees)')
timed(d, 'D again', steps=30)
