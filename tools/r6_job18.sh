#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6i_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6i_tests.log
timeout 400 python tools/pool_scale.py --workers 1,4,8,12 --seconds 5 --modes pipeline > gpurun_out/r6i_pool_scale.json 2> gpurun_out/r6i_pool_scale.err; grep '^pipeline' gpurun_out/r6i_pool_scale.err | cut -c1-300
timeout 400 python tools/pool_scale.py --workers 1,4,8,12 --seconds 5 --modes pipeline --no-poisson > gpurun_out/r6i_pool_scale_np.json 2> gpurun_out/r6i_pool_scale_np.err; grep '^pipeline' gpurun_out/r6i_pool_scale_np.err | cut -c1-300
VKX_SYNC=block timeout 400 python tools/pool_scale.py --workers 12,16 --seconds 5 --modes pipeline 2>&1 | grep '^pipeline' | cut -c1-300
./tools/probes/page_dispatches.sh r6i 40 > /dev/null
./tools/pool_trace.sh r6i 8 --no-poisson
