#!/bin/bash
# dispatches per page by C entry point AND kernel -> gpurun_out/<tag>_page_dispatches.txt     usage: page_dispatches.sh <tag> [pages]
# Two passes of the same deterministic page sequence: one with --kernel-rename (every dispatch carries the ROCTx range = entry point that issued it),
# one without (the real kernel names); the two dispatch sequences are zipped in stream order.
TAG=${1:-r6}; PAGES=${2:-40}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd /tmp/pd2
timeout 300 rocprofv3 --kernel-trace --marker-trace --kernel-rename --output-format csv -d /tmp/pd -o t -- python $ROOT/tools/probes/page_dispatches.py $PAGES > /tmp/pd.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pd2 -o t -- python $ROOT/tools/probes/page_dispatches.py $PAGES > /tmp/pd2.log 2>&1
python - $PAGES > $ROOT/gpurun_out/${TAG}_page_dispatches.txt <<'PY'
import csv, glob, sys, collections
pages = int(sys.argv[1])
def load(root):
    rows = []
    for p in glob.glob(root + '/**/*kernel_trace.csv', recursive=True):
        rows += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(p))]
    rows.sort()
    return rows
a, b = load('/tmp/pd'), load('/tmp/pd2')
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].split('<')[0]
zipped = len(a) == len(b)
n = collections.Counter(); t = collections.Counter(); detail = collections.defaultdict(collections.Counter)
for i, (s, e, name) in enumerate(a):
    if name.startswith('warmup_') or ':' not in name:
        continue
    n[name] += 1; t[name] += e - s
    if zipped:
        detail[name][short(b[i][2])] += 1
tot = sum(n.values())
print(f'dispatches per page: {tot / pages:.1f}   (kernel time per page {sum(t.values()) / pages / 1e3:.0f} us; sequences zipped: {zipped}, {len(a)} vs {len(b)})')
for step in ('assembler', 'distortion', 'resizing'):
    rows = [(k, v) for k, v in n.items() if k.startswith(step)]
    print(f'{step}: {sum(v for _, v in rows) / pages:.1f}')
    for k, v in sorted(rows, key=lambda kv: -kv[1]):
        d = ', '.join(f'{kn} {c / pages:.2f}' for kn, c in detail[k].most_common(8))
        print(f'   {k:52s} {v / pages:6.2f} dispatches {t[k] / pages / 1e3:7.1f} us   [{d}]')
PY
cat $ROOT/gpurun_out/${TAG}_page_dispatches.txt
