import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from numpy.random import default_rng
from vkit_amd import _native as N
from vkit_amd.hostpipe import HostPipeline
from vkit_amd.mechanism import distortion as D
from vkit_amd.mechanism.distortion_policy.geometric import mls as P_mls
ctx = N.default_ctx()
S = 2048
gen = P_mls.SimilarityMlsConfigGenerator(P_mls.SimilarityMlsConfigGeneratorConfig(), 5)
sts = [D.similarity_mls.generate_state(gen((S, S), default_rng(i)), (S, S)) for i in range(4)]
imgs = []
for i in range(4):
    img = ctx.pinned_empty((S, S, 3), np.uint8); img[...] = 7 + i; imgs.append(img)
for lanes, depth in ((8, 8), (8, 16), (12, 24)):
    with HostPipeline(ctx, depth=depth, lanes=lanes) as pipe:
        for k in range(depth): pipe.submit_remap([imgs[k % 4]], sts[k % 4])
        pipe.drain()
        t0 = time.perf_counter(); tickets = []; n = 300
        tsub = 0.0
        for k in range(n):
            a = time.perf_counter()
            tickets.append(pipe.submit_remap([imgs[k % 4]], sts[k % 4]))
            tsub += time.perf_counter() - a
            if k >= depth - 1: pipe.result(tickets[k - depth + 1])
        pipe.drain()
        dt = time.perf_counter() - t0
    print(f'lanes {lanes} depth {depth}: {dt / n * 1e3:.3f} ms/job = {n * S * S / dt / 1e9:.2f} Gpx/s; submit call {tsub / n * 1e3:.3f} ms')
