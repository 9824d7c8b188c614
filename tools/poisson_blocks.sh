#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
OUT=gpurun_out/r6_poisson_blocks.txt; : > $OUT
for envs in "X=1" "VKX_PZ_BLOCKS=192" "VKX_PZ_BLOCKS=128" "VKX_PZ_BLOCKS=96" "VKX_PZ_BLOCKS=64" "VKX_PZ_BLOCKS=128 VKX_PZ_DEPTH=3" "VKX_PZ_BLOCKS=64 VKX_PZ_DEPTH=3"; do
  echo "== $envs" >> $OUT
  env $envs VKX_PZ_PROBE=1 timeout 200 python tools/poisson_probe.py 1024 2> /tmp/pz.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"case\"'):
        r=json.loads(l)
        if r['case'] in ('uniform bytes','page','all 255','dark 0..12','gray plane'):
            print('  %-14s device %.2f ms  equal %s stream %s %s' % (r['case'], r['device_ms'], r.get('equal'), r.get('stream_equal'), r.get('kernels_ms',{}).get('k_pz_super')))
    if 'EQUAL' in l or 'MISMATCH' in l: print(' ', l.strip())
" >> $OUT
  grep "pz probe" /tmp/pz.err | sed -n 2,2p | cut -c1-260 >> $OUT
done
cat $OUT
