#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
for mode in spin block yield; do
  echo "== VKX_SYNC=$mode"
  VKX_SYNC=$mode timeout 300 python tools/pool_scale.py --workers 1,8,12,16 --seconds 4 --modes pipeline > gpurun_out/r6e_pool_$mode.json 2> gpurun_out/r6e_pool_$mode.err; grep '^pipeline' gpurun_out/r6e_pool_$mode.err | cut -c1-330
done
echo "== mapped layers, 1 worker"
VKX_LAYERS_MAPPED=1 timeout 300 python tools/pool_scale.py --workers 1 --seconds 4 --modes pipeline 2>&1 | grep '^pipeline' | cut -c1-330
