#!/usr/bin/env python3
"""Prints per-wavefront instruction counts and activity of k_chain_fused from gpurun_out/pmc_<tag> (tools/pmc_quick.sh)."""
import csv, glob, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
acc = collections.defaultdict(list)
for path in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', f'pmc_{tag}', 'p*', '**', '*counter_collection.csv'), recursive=True)):
    for r in csv.DictReader(open(path)):
        if 'k_chain_fused' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            acc['ns_' + os.path.basename(os.path.dirname(path))[:2]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m.get('SQ_WAVES', 1)
for k in sorted(m):
    print(f'{k:24s} {m[k]:14.6g}  per wave {m[k] / w:10.2f}')
