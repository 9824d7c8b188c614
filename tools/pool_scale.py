#!/usr/bin/env python3
"""Pages per second through the reference's API on ONE GPU, scaled the way the reference scales: processes
(vkit/utility/pool.py:153-243, vkit/pipeline/pool.py:64-118 -- a pool of worker processes, each running whole pipeline runs with
its own rng).  k = 1, 2, 4, ... workers share GPU 0; every worker runs PageAssemblerStep -> PageDistortionStep -> PageResizingStep
on C4-shaped pages (1024^2, 64 text lines, 384 char polygons; resized outputs read on the host), page after page with its own
seeds, between a common start barrier and a fixed number of pages.  Also: RandomDistortion.distort on 1024^2 images.
Reports pages/s, the per-page latency inside a worker, and the GPU busy share (sum of kernel time by HIP events / wall).
Usage: tools/pool_scale.py [--workers 1,2,4,8,16,32] [--pages 40] [--no-poisson] > out.json"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, n_workers, pages, mode, no_poisson, barrier, queue):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np
    from numpy.random import default_rng
    from vkit_amd import _native as N
    ctx = N.default_ctx()
    if mode == 'pipeline':
        from test_gpu_composite import _synthetic_page_input
        from vkit_amd.pipeline import text_detection as T
        step_input = _synthetic_page_input(seed=3 + rank, size=1024, n_lines=64)
        assembler = T.page_assembler_step_factory.create()
        dconf = {'random_distortion_factory_config': {'disabled_policy_names': ['defocus_blur', 'zoom_in_blur', 'poisson_noise']}} if no_poisson else None
        distortion = T.page_distortion_step_factory.create(dconf)
        resizing = T.page_resizing_step_factory.create()

        def page(seed):
            rng = default_rng(seed)
            a = assembler.run(step_input, rng)
            d = distortion.run(T.PageDistortionStepInput(a), rng)
            r = resizing.run(T.PageResizingStepInput(d), rng)
            return int(r.page_image.mat[0, 0, 0]) + int(r.page_char_mask.mat.sum() > 0)
    else:
        from vkit_amd.element import Image
        from vkit_amd.mechanism.distortion_policy import random_distortion_factory
        rd = random_distortion_factory.create({'disabled_policy_names': ['poisson_noise']} if no_poisson else None)
        img = Image(mat=default_rng(100 + rank).integers(0, 256, (1024, 1024, 3), dtype=np.uint8))

        def page(seed):
            return int(rd.distort(default_rng(seed), image=img).image.mat[0, 0, 0])
    for s in range(3):
        page(10_000 * rank + s)          # warm up: library, pools, tables
    ctx.sync()
    if os.environ.get('VKX_POOL_TIMING', '1') != '0':      # 0: no per-launch events (gpu_busy_share reads 0): the pages/s of an uninstrumented worker
        ctx.set_timing(True)
    ctx.reset_timings()
    barrier.wait()
    t0 = time.time()
    lat = []
    for s in range(pages):
        t1 = time.perf_counter()
        page(10_000 * rank + 100 + s)
        lat.append(time.perf_counter() - t1)
    ctx.sync()
    t1 = time.time()
    gpu_ms = sum(v[0] for v in ctx.timings().values())
    queue.put((rank, t0, t1, sorted(lat), gpu_ms))


def run(n_workers, pages, mode, no_poisson):
    ctx = mp.get_context('spawn')
    barrier, queue = ctx.Barrier(n_workers), ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, n_workers, pages, mode, no_poisson, barrier, queue)) for r in range(n_workers)]
    for p in procs:
        p.start()
    got, deadline = [], time.time() + 600
    while len(got) < n_workers and time.time() < deadline:
        try:
            got.append(queue.get(timeout=1.0))
        except Exception:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    for p in procs:
        p.join(timeout=10)
        if p.is_alive():
            p.terminate()
    if len(got) < n_workers:
        return {'workers': n_workers, 'error': 'a worker failed'}
    wall = max(g[2] for g in got) - min(g[1] for g in got)
    lat = sorted(x for g in got for x in g[3])
    return {'workers': n_workers, 'pages': n_workers * pages, 'wall_s': round(wall, 3), 'pages_per_s': round(n_workers * pages / wall, 1),
            'latency_ms': {'mean': round(sum(lat) / len(lat) * 1e3, 2), 'median': round(lat[len(lat) // 2] * 1e3, 2),
                           'p90': round(lat[int(len(lat) * 0.9)] * 1e3, 2), 'max': round(lat[-1] * 1e3, 2)},
            'gpu_busy_share': round(sum(g[4] for g in got) / 1e3 / wall, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workers', default='1,2,4,8,16,32')
    ap.add_argument('--pages', type=int, default=40)
    ap.add_argument('--no-poisson', action='store_true', help='poisson_noise (a sequential numpy call of ~95 ms per 1024^2 page) disabled')
    args = ap.parse_args()
    out = {'host_cores': os.cpu_count(), 'pages_per_worker': args.pages, 'poisson_noise': not args.no_poisson,
           'page': '1024x1024, 64 text lines, 384 char polygons (tests/test_gpu_composite.py::_synthetic_page_input)'}
    for mode in ('pipeline', 'random_distortion'):
        out[mode] = [run(k, args.pages, mode, args.no_poisson) for k in (int(w) for w in args.workers.split(','))]
        print(mode, json.dumps(out[mode]), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
