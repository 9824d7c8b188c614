#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6c_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r6c_tests.log
./tools/probes/page_dispatches.sh r6c 40 | head -60
timeout 300 python tools/pool_scale.py --workers 1,4,8 --seconds 4 --modes pipeline > gpurun_out/r6c_pool_scale.json 2> gpurun_out/r6c_pool_scale.err; grep '^pipeline' gpurun_out/r6c_pool_scale.err | cut -c1-330
timeout 300 python tools/pool_scale.py --workers 1,8 --seconds 4 --modes pipeline --no-poisson > gpurun_out/r6c_pool_scale_np.json 2> gpurun_out/r6c_pool_scale_np.err; grep '^pipeline' gpurun_out/r6c_pool_scale_np.err | cut -c1-330
