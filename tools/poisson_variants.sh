#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
OUT=gpurun_out/r6_poisson_variants.txt; : > $OUT
for envs in "X=1" "VKX_PZ_DEPTH=1" "VKX_PZ_DEPTH=2" "VKX_PZ_DEPTH=3" "VKX_PZ_GRID=128" "VKX_PZ_GRID=192" "VKX_PZ_RESOLVE_ROWS=6" "VKX_PZ_RESOLVE_ROWS=20" "VKX_PZ_G_GLOBAL=1" "VKX_PZ_ORDER=0"; do
  echo "== $envs" >> $OUT
  env $envs VKX_PZ_PROBE=1 timeout 200 python tools/poisson_probe.py 1024 2> /tmp/pz.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"case\"'):
        r=json.loads(l)
        if r['case'] in ('uniform bytes','page','all 255','dark 0..12'):
            print('  %-14s device %.2f ms  equal %s  %s' % (r['case'], r['device_ms'], r.get('equal'), r.get('kernels_ms',{}).get('k_pz_super')))
" >> $OUT
  grep "pz probe" /tmp/pz.err | sed -n 2,4p | cut -c1-260 >> $OUT
done
cat $OUT
