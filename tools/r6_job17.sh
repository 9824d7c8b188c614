#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
for v in 0 1; do
  echo "== VKX_STAGE_COPY_STREAM=$v"
  VKX_STAGE_COPY_STREAM=$v timeout 300 python tools/pool_scale.py --workers 1,8,12 --seconds 4 --modes pipeline --no-poisson 2>&1 | grep '^pipeline' | cut -c1-250
  VKX_STAGE_COPY_STREAM=$v ./tools/probes/page_dispatches.sh r6h$v 40 | sed -n 1,6p
done
VKX_STAGE_COPY_STREAM=1 timeout 300 python -m pytest tests/test_gpu_page_dispatch_diet.py tests/test_gpu_composite.py -x -q -m gpu 2>&1 | tail -2
