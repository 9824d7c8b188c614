#!/bin/bash
# dispatches per host <-> device operation: one rocprofv3 run per op kind -> gpurun_out/r6_copy_kinds.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_copy_kinds.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
for op in none upload_3mb_pageable upload_3mb_pinned upload_1mb_pageable upload_64kb_pageable upload_256b_pageable download_3mb_pageable download_3mb_pinned download_4b_pageable memset_1mb memset_4b d2d_3mb to_device_3mb; do
  rm -rf /tmp/ck; 
  timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ck -o t -- python $ROOT/tools/probes/copy_kinds.py $op 20 > /tmp/ck.log 2>&1
  python - "$op" >> $OUT <<'PY'
import csv, glob, sys, collections
op = sys.argv[1]
k = collections.Counter(); kt = collections.Counter()
for p in glob.glob('/tmp/ck/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        n = r['Kernel_Name'].split('(')[0]
        k[n] += 1; kt[n] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
c = collections.Counter(); ct = collections.Counter()
for p in glob.glob('/tmp/ck/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        c[r['Direction']] += 1; ct[r['Direction']] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
host = [l for l in open('/tmp/ck.log') if 'host us per rep' in l]
print(f'{op:26s} per 20 reps: kernels', {n: (v, round(kt[n] / v / 1e3, 1)) for n, v in k.items()}, 'copies', {n: (v, round(ct[n] / v / 1e3, 1)) for n, v in c.items()}, (host[0].strip() if host else ''))
PY
done
cat $OUT
