#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
OUT=gpurun_out/r6_poisson_grid_pool.txt; : > $OUT
for g in 256 0; do
  echo "== VKX_PZ_GRID=$g (0 = the default: 7/8 of the CUs)" >> $OUT
  VKX_PZ_GRID=$g python tools/pool_scale.py --workers 1,4,8 --seconds 5 --modes pipeline 2>&1 | grep "^pipeline" | cut -c1-330 >> $OUT
  VKX_PZ_GRID=$g python tools/poisson_probe.py 1024 2>/dev/null | tail -1 | cut -c1-700 >> $OUT
done
cat $OUT
timeout 600 python -m pytest tests/test_gpu_poisson.py tests/test_gpu_soak.py -q -x -m gpu 2>&1 | tail -2
