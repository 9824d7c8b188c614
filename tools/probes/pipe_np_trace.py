"""24 numpy-stream chain jobs through an 8-deep HostPipeline (for a rocprofv3 --kernel-trace --memory-copy-trace timeline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from numpy.random import default_rng
from vkit_amd import _native as N
from vkit_amd.hostpipe import HostPipeline
from vkit_amd.mechanism.distortion_policy.geometric import mls as P_mls
from vkit_amd.mechanism import distortion as D
mode = sys.argv[1] if len(sys.argv) > 1 else 'numpy_stream'
ctx = N.default_ctx()
S = 2048
n_img = 8
rng = default_rng(0)
pinned = []
for i in range(n_img):
    a = ctx.pinned_empty((S, S, 3), np.uint8)
    a[...] = rng.integers(0, 256, (S, S, 3), dtype=np.uint8)
    pinned.append(a)
gen = P_mls.SimilarityMlsConfigGenerator(P_mls.SimilarityMlsConfigGeneratorConfig(), 5)
states = [D.similarity_mls.generate_state(gen((S, S), default_rng(i)), (S, S)) for i in range(n_img)]
pipe = HostPipeline(ctx, depth=int(os.environ.get('DEPTH', '8')), lanes=int(os.environ.get('LANES', '8')))
if os.environ.get('TIMING'):
    for lane in pipe.lanes:
        lane.set_timing(True)
extra = int(os.environ.get('EXTRA', '0'))
def submit(i):
    if extra:
        slot = pipe.slots[pipe._next % len(pipe.slots)]
        N.lib().vkx_dbg_spin(slot.ctx.handle, extra, 2000)
    if mode == 'device_noise':
        return pipe.submit_chain(pinned[i % n_img], states[i % n_img], blur_sigma=1.2, hue_delta=7, noise_std=10.0, noise_seed=i)
    return pipe.submit_chain(pinned[i % n_img], states[i % n_img], blur_sigma=1.2, hue_delta=7, noise_std=10.0, noise_rng=default_rng(i))
for rep in range(2):
    t0 = time.perf_counter()
    tickets = []
    for i in range(24):
        tickets.append(submit(i))
        if len(tickets) >= len(pipe.slots):
            pipe.result(tickets.pop(0))
    for t in tickets:
        pipe.result(t)
    print(mode, 'ms/image', (time.perf_counter() - t0) / 24 * 1e3)
if os.environ.get('TIMING'):
    for k, lane in enumerate(pipe.lanes[:3]):
        print(k, {n: (round(v[0], 3), v[1]) for n, v in lane.timings().items()})
pipe.close()
