#!/usr/bin/env python3
"""One worker of tools/pool_scale.py as a process of its own (so that it can run under its own rocprofv3: the children of a
multiprocessing pool die in the profiler's finalizer and leave no trace).  The workers of a pool meet at a directory barrier:
    tools/pool_worker.py <rank> <n_workers> <seconds> <barrier dir> [--no-poisson]
Prints one JSON line: rank, pages, wall, cpu seconds, latencies."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
    rank, n, seconds, bdir = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
    no_poisson = '--no-poisson' in sys.argv
    from pool_scale import _make_page_fn
    from vkit_amd import _native as N
    ctx = N.default_ctx()
    page = _make_page_fn(rank, 'pipeline', no_poisson)
    for s in range(3):
        page(10_000 * rank + s)
    ctx.sync()
    open(os.path.join(bdir, f'ready{rank}'), 'w').close()
    deadline = time.time() + 300
    while len([f for f in os.listdir(bdir) if f.startswith('ready')]) < n and time.time() < deadline:
        time.sleep(0.002)
    t0, c0 = time.time(), time.process_time()
    lat, s = [], 0
    while time.time() - t0 < seconds:
        t1 = time.perf_counter()
        page(10_000 * rank + 100 + s)
        lat.append(time.perf_counter() - t1)
        s += 1
    ctx.sync()
    t1, c1 = time.time(), time.process_time()
    print(json.dumps({'rank': rank, 'pages': s, 't0': t0, 't1': t1, 'cpu_s': c1 - c0, 'mean_ms': sum(lat) / len(lat) * 1e3}), flush=True)


if __name__ == '__main__':
    main()
