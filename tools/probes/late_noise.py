"""Would adding the noise in the generator's placement pass (NP_NORMAL_ADD_U8 in place on the chain's output) beat the
chain reading an int16 plane?  64 images of the bench workload, kernel times by HIP events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from numpy.random import default_rng
from vkit_amd import _native as N
from vkit_amd.batch import ChainBatch
from vkit_amd.mechanism import distortion as D
from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam

ctx = N.default_ctx()
S, B = 2048, 64
rng = default_rng(0)
img = rng.integers(0, 256, (S, S, 3), dtype=np.uint8)
cgen = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), 5)
states = [D.camera_cubic_curve.generate_state(cgen((S, S), default_rng(i)), (S, S)) for i in range(8)]

def run(mode):
    batch = ChainBatch(ctx)
    for j in range(B):
        if mode == 'plane':
            batch.add(img, states[j % 8], blur_sigma=1.0, hue_delta=37, noise_std=10.0, noise_rng=default_rng(5000 + j))
        else:
            batch.add(img, states[j % 8], blur_sigma=1.0, hue_delta=37)
    jobs = res = None
    if mode == 'late':
        jobs = (N.VkxNpJob * B)()
        for j in range(B):
            dh, dw = batch._dst_shapes[j]
            jobs[j] = N.np_job(N.NP_NORMAL_ADD_U8, N.np_stream(default_rng(5000 + j)), dh * dw * 3, 10.0, src=batch._items[j].dst, dst=batch._items[j].dst)
        res = N.NpResults(ctx, B)
    def step():
        batch.run()
        if mode == 'late':
            N.check(N.lib().vkx_np_draw_batch_dev(ctx.handle, jobs, B, res.array))
    step(); ctx.sync()
    ctx.set_timing(True); ctx.reset_timings()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    ctx.sync()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    k = {n: round(v[0] / 5, 3) for n, v in ctx.timings().items()}
    ctx.set_timing(False)
    out = batch.result(3)
    batch.close()
    return dt, k, out

dt_p, k_p, out_p = run('plane')
dt_l, k_l, out_l = run('late')
print('plane mode  ms/step', round(dt_p, 3), k_p)
print('late  mode  ms/step', round(dt_l, 3), k_l)
print('identical result', bool((out_p == out_l).all()))
