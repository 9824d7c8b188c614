#!/usr/bin/env python3
"""Per-wavefront instruction counts and activity of the k_np_* kernels from gpurun_out/pmc_<tag> (tools/np_pmc.sh)."""
import csv, glob, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', f'pmc_{tag}', 'p*', '**', '*counter_collection.csv'), recursive=True)):
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name']
        for key in ('k_np_draw', 'k_np_place<', 'k_np_resolve', 'k_np_fused', 'k_np_apply', 'k_np_place_walk', 'k_np_tiles_finish'):
            if key in name:
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
                acc[key]['ns'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for key, d in acc.items():
    m = {k: sum(v) / len(v) for k, v in d.items()}
    w = m.get('SQ_WAVES', 1)
    print(f'== {key}  (waves {w:.0f}, {m["ns"] / 1e3:.1f} us under the profiler)')
    for k in sorted(m):
        print(f'  {k:24s} {m[k]:14.6g}  per wave {m[k] / w:10.2f}')
