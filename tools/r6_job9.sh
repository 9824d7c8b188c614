#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_composite.py tests/test_gpu_operators.py tests/test_gpu_polygon.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r6d_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6d_tests.log
timeout 300 python tools/probes/page_profile.py 96 2>&1 | sed -n 3,4p
./tools/probes/page_dispatches.sh r6d 40 | head -12
timeout 300 python tools/pool_scale.py --workers 1,4,8 --seconds 4 --modes pipeline > gpurun_out/r6d_pool_scale.json 2> gpurun_out/r6d_pool_scale.err; grep '^pipeline' gpurun_out/r6d_pool_scale.err | cut -c1-330
timeout 300 python tools/pool_scale.py --workers 1,8,12 --seconds 4 --modes pipeline --no-poisson > gpurun_out/r6d_pool_scale_np.json 2> gpurun_out/r6d_pool_scale_np.err; grep '^pipeline' gpurun_out/r6d_pool_scale_np.err | cut -c1-330
