#!/usr/bin/env python3
"""Timeline of one step from a rocprofv3 --kernel-trace csv: tools/timeline.py <kernel_trace.csv> [step index from the end]
Prints every dispatch of the step with start / end relative to the step's first kernel, its stream (queue) and the gap to the
previous end on the same queue, then the busy / idle summary."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].split('<')[0],
              str(r.get('Stream_Id', r.get('Queue_Id', '?')))) for r in rows), key=lambda e: e[0])
# a step starts at a k_np_tile_states dispatch that follows a k_chain_fused
marks = []
last_fused = -1
for i, e in enumerate(ev):
    if 'k_chain_fused' in e[2]:
        last_fused = i
    if 'k_np_tile_states' in e[2] and last_fused >= 0 and all('k_chain_fused' not in x[2] for x in ev[last_fused + 1:i]) and (not marks or marks[-1] < last_fused):
        marks.append(i)
if len(marks) < back + 1:
    raise SystemExit(f'only {len(marks)} step boundaries found')
a, b = marks[-back - 1], marks[-back]
step = ev[a:b]
t0 = step[0][0]
last_end = {}
print(f'{"kernel":28s} {"queue":>6s} {"start us":>10s} {"end us":>10s} {"dur us":>9s} {"gap on queue":>12s}')
for s, e, name, q in step:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print(f'{name[-28:]:28s} {q:>6s} {(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {gap:12.1f}')
# union of busy intervals
iv = sorted((s, e) for s, e, _, _ in step)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = ev[b][0] - t0
print(f'step span {span / 1e3:.1f} us, some kernel running {busy / 1e3:.1f} us, nothing running {(span - busy) / 1e3:.1f} us')
