#!/usr/bin/env python3
"""Pages per second through the reference's API on ONE GPU, scaled the way the reference scales: processes
(vkit/utility/pool.py:153-243, vkit/pipeline/pool.py:64-118 -- a pool of worker processes, each running whole pipeline runs with
its own rng) -- and, beside them, threads of one process (one context = stream + scratch per thread, _native.default_ctx()).
k = 1, 2, 4, ... workers share GPU 0; every worker runs PageAssemblerStep -> PageDistortionStep -> PageResizingStep on C4-shaped
pages (1024^2, 64 text lines, 384 char polygons; resized outputs read on the host), page after page with its own seeds, from a
common start barrier for a fixed number of SECONDS (round 6: >= 5 s per point; rounds 4 - 5 ran 40 pages = 0.2 - 0.6 s).
Also: RandomDistortion.distort on 1024^2 images.

Per point: pages/s, the per-page latency inside a worker (mean / median / p90 / max), and what bounds it --
  cpu_s_per_page      user + system time of the worker per page (os.times): equal to the wall time per page when the worker is
                      bound by its own Python, smaller when it waits (for the GPU, or for a CPU to run on)
  cpus_busy           busy CPUs of the whole box during the point (/proc/stat deltas)
  throttled_ms        time the cgroup's CPU quota held its tasks back during the point (cpu.stat: throttled_usec)
  gpu_busy_share      kernel time / wall: from per-launch HIP events when VKX_POOL_TIMING=1 (they cost the workers themselves), from
                      the rocprofv3 --kernel-trace of the whole pool otherwise (tools/pool_trace.sh)
Usage: tools/pool_scale.py [--workers 1,2,4,8] [--seconds 5] [--kind processes,threads] [--modes pipeline,random_distortion]
                           [--no-poisson] > out.json"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _proc_stat():
    line = (_read('/proc/stat') or 'cpu 0 0 0 0').splitlines()[0].split()[1:]
    vals = [int(v) for v in line]
    idle = vals[3] + (vals[4] if len(vals) > 4 else 0)
    return sum(vals), idle


def _throttled_usec():
    text = _read('/sys/fs/cgroup/cpu.stat') or _read('/sys/fs/cgroup/cpu/cpu.stat') or ''
    out = {}
    for ln in text.splitlines():
        k, _, v = ln.partition(' ')
        out[k] = int(v) if v.strip().isdigit() else 0
    if 'throttled_usec' in out:
        return out['throttled_usec'], out.get('nr_throttled', 0)
    if 'throttled_time' in out:                 # cgroup v1: nanoseconds
        return out['throttled_time'] // 1000, out.get('nr_throttled', 0)
    return None, None


def _make_page_fn(rank, mode, no_poisson):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np
    from numpy.random import default_rng
    if mode == 'pipeline':
        from vkit_amd.pipeline.text_detection.synthetic_page import synthetic_page_input as _synthetic_page_input
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
    return page


def _worker(rank, seconds, mode, no_poisson, barrier, put, thread_cpu):
    from vkit_amd import _native as N
    ctx = N.default_ctx()
    page = _make_page_fn(rank, mode, no_poisson)
    for s in range(3):
        page(10_000 * rank + s)          # warm up: library, pools, tables
    ctx.sync()
    timing = os.environ.get('VKX_POOL_TIMING', '0') != '0'     # per-launch events (they cost the worker itself): off by default since round 6
    if timing:
        ctx.set_timing(True)
    ctx.reset_timings()
    cpu_clock = time.thread_time if thread_cpu else time.process_time
    barrier.wait()
    t0, c0 = time.time(), cpu_clock()
    lat, s = [], 0
    while True:
        t1 = time.perf_counter()
        page(10_000 * rank + 100 + s)
        lat.append(time.perf_counter() - t1)
        s += 1
        if time.time() - t0 >= seconds:
            break
    ctx.sync()
    t1, c1 = time.time(), cpu_clock()
    gpu_ms = sum(v[0] for v in ctx.timings().values()) if timing else None
    put((rank, t0, t1, sorted(lat), gpu_ms, c1 - c0))


def run(n_workers, seconds, mode, no_poisson, kind):
    results = []
    if kind == 'processes':
        ctx = mp.get_context('spawn')
        barrier, queue = ctx.Barrier(n_workers + 1), ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, seconds, mode, no_poisson, barrier, queue.put, False)) for r in range(n_workers)]
        alive = lambda: not any(p.exitcode not in (None, 0) for p in procs)   # noqa: E731
        get = lambda: queue.get(timeout=1.0)                                  # noqa: E731
    else:
        import queue as Q
        barrier, q = threading.Barrier(n_workers + 1), Q.Queue()
        procs = [threading.Thread(target=_worker, args=(r, seconds, mode, no_poisson, barrier, q.put, True), daemon=True) for r in range(n_workers)]
        alive = lambda: True                                                  # noqa: E731
        get = lambda: q.get(timeout=1.0)                                      # noqa: E731
    for p in procs:
        p.start()
    barrier.wait(timeout=600)
    stat0, thr0 = _proc_stat(), _throttled_usec()
    deadline = time.time() + seconds + 600
    while len(results) < n_workers and time.time() < deadline:
        try:
            results.append(get())
        except Exception:
            if not alive():
                break
    stat1, thr1 = _proc_stat(), _throttled_usec()
    for p in procs:
        p.join(timeout=10)
        if kind == 'processes' and p.is_alive():
            p.terminate()
    if len(results) < n_workers:
        return {'workers': n_workers, 'kind': kind, 'error': 'a worker failed'}
    wall = max(g[2] for g in results) - min(g[1] for g in results)
    lat = sorted(x for g in results for x in g[3])
    pages = len(lat)
    cpu_s = sum(g[5] for g in results)
    hz = os.sysconf('SC_CLK_TCK')
    out = {'workers': n_workers, 'kind': kind, 'pages': pages, 'wall_s': round(wall, 3), 'pages_per_s': round(pages / wall, 1),
           'latency_ms': {'mean': round(sum(lat) / len(lat) * 1e3, 2), 'median': round(lat[len(lat) // 2] * 1e3, 2),
                          'p90': round(lat[int(len(lat) * 0.9)] * 1e3, 2), 'max': round(lat[-1] * 1e3, 2)},
           'cpu_ms_per_page': round(cpu_s / pages * 1e3, 2), 'wall_ms_per_page_in_worker': round(sum(lat) / pages * 1e3, 2),
           'cpus_busy': round(((stat1[0] - stat0[0]) - (stat1[1] - stat0[1])) / hz / max(wall, 1e-9), 2)}
    if thr0[0] is not None:
        out['throttled_ms'] = round((thr1[0] - thr0[0]) / 1e3, 1)
        out['nr_throttled'] = thr1[1] - thr0[1]
    gpu = [g[4] for g in results if g[4] is not None]
    if gpu:
        out['gpu_busy_share'] = round(sum(gpu) / 1e3 / wall, 3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workers', default='1,2,4,8')
    ap.add_argument('--seconds', type=float, default=5.0)
    ap.add_argument('--kind', default='processes', help='processes, threads or both (comma separated)')
    ap.add_argument('--modes', default='pipeline,random_distortion')
    ap.add_argument('--no-poisson', action='store_true', help='poisson_noise disabled')
    args = ap.parse_args()
    sys.path.insert(0, ROOT)
    out = {'host_cores': os.cpu_count(), 'affinity': len(os.sched_getaffinity(0)), 'cgroup_cpu_max': (_read('/sys/fs/cgroup/cpu.max') or '').strip() or None,
           'seconds_per_point': args.seconds, 'poisson_noise': not args.no_poisson,
           'page': '1024x1024, 64 text lines, 384 char polygons (vkit_amd/pipeline/text_detection/synthetic_page.py)'}
    for mode in args.modes.split(','):
        for kind in args.kind.split(','):
            key = mode if kind == 'processes' else f'{mode}_{kind}'
            out[key] = []
            for k in (int(w) for w in args.workers.split(',')):
                out[key].append(run(k, args.seconds, mode, args.no_poisson, kind))
                print(key, json.dumps(out[key][-1]), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
