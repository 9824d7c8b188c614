#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
OUT=gpurun_out/r6_env_variants.txt; : > $OUT
for envs in "X=1" "HSA_ENABLE_SDMA=0" "HSA_ENABLE_INTERRUPT=0" "HSA_ENABLE_SDMA=0 HSA_ENABLE_INTERRUPT=0"; do
  echo "== $envs" >> $OUT
  for op in upload_3mb_pageable upload_3mb_pinned upload_64kb_pageable upload_256b_pageable download_3mb_pinned memset_1mb d2d_3mb; do
    env $envs python tools/probes/copy_kinds.py $op 40 2>/dev/null | tail -1 >> $OUT
  done
  env $envs timeout 300 python tools/pool_scale.py --workers 1,8 --seconds 3 --modes pipeline > /tmp/p.json 2> /tmp/p.err
  grep '^pipeline' /tmp/p.err | cut -c1-420 >> $OUT
done
cat $OUT
