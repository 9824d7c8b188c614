#!/usr/bin/env python3
"""GPU side of a pool of workers from the rocprofv3 traces of the whole pool (every worker process writes its own files):
    tools/pool_trace.py <output dir of rocprofv3 --kernel-trace --memory-copy-trace> [pages]
Kernel (and copy) intervals of ALL processes on one time axis: the share of the wall during which some kernel ran (gpu_busy_share),
how much of it more than one kernel ran (overlap between processes / streams), time per kernel name, dispatches per page, and
the gaps."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def union(iv):
    iv = sorted(iv)
    if not iv:
        return 0, 0
    busy, multi = 0, 0
    cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            multi += min(e, ce) - s
            ce = max(ce, e)
    busy += ce - cs
    return busy, multi


def main():
    root = sys.argv[1]
    pages = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    kern, copies = [], []
    per_pid = defaultdict(int)
    for path in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].split('<')[0]
            kern.append((s, e, name, path))
            per_pid[path] += 1
    for path in glob.glob(os.path.join(root, '**', '*memory_copy_trace.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            copies.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Direction', '?')))
    if not kern:
        raise SystemExit('no kernel trace found under ' + root)
    # the steady window: from the latest first-kernel of any process to the earliest last-kernel (the warm-up of a late starter and the
    # tail of an early finisher are not the pool at full strength); processes with few dispatches (the parent) are ignored
    big = [p for p, n in per_pid.items() if n >= 0.2 * max(per_pid.values())]
    t_lo = max(min(k[0] for k in kern if k[3] == p) for p in big)
    t_hi = min(max(k[1] for k in kern if k[3] == p) for p in big)
    # drop the first 20 % (warm-up pages run before the barrier)
    t_lo = t_lo + (t_hi - t_lo) // 5
    win = [(max(s, t_lo), min(e, t_hi), n) for s, e, n, _ in kern if e > t_lo and s < t_hi]
    wall = t_hi - t_lo
    busy, multi = union([(s, e) for s, e, _ in win])
    by_name = defaultdict(lambda: [0, 0])
    for s, e, n in win:
        by_name[n][0] += e - s
        by_name[n][1] += 1
    cwin = [(max(s, t_lo), min(e, t_hi), d) for s, e, d in copies if e > t_lo and s < t_hi]
    cbusy, _ = union([(s, e) for s, e, _ in cwin])
    cdir = defaultdict(lambda: [0, 0])
    for s, e, d in cwin:
        cdir[d][0] += e - s
        cdir[d][1] += 1
    out = {'processes': len(big), 'window_s': wall / 1e9, 'gpu_busy_share': busy / wall, 'kernel_time_sum_over_wall': sum(e - s for s, e, _ in win) / wall,
           'overlapped_share': multi / wall, 'dispatches_per_s': len(win) / (wall / 1e9), 'copy_busy_share': cbusy / wall,
           'copies_per_s': len(cwin) / (wall / 1e9),
           'copies': {d: {'ms': v[0] / 1e6, 'n': v[1]} for d, v in cdir.items()},
           'top_kernels': [{'kernel': n, 'share_of_wall': v[0] / wall, 'dispatches': v[1], 'avg_us': v[0] / v[1] / 1e3}
                           for n, v in sorted(by_name.items(), key=lambda kv: -kv[1][0])[:25]]}
    if pages:
        out['pages_in_window_estimate'] = pages
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
