#!/bin/bash
# End-of-round check on the GPU box: the whole -m gpu suite, then the worker pool with and without poisson_noise (5 s per point)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/final_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/final_tests.log
timeout 400 python tools/pool_scale.py --workers 1,4,8 --seconds 5 --modes pipeline > gpurun_out/r6n_pool_scale.json 2> gpurun_out/r6n_pool_scale.err; grep '^pipeline' gpurun_out/r6n_pool_scale.err | cut -c1-260
timeout 400 python tools/pool_scale.py --workers 1,4,8,12 --seconds 5 --modes pipeline --no-poisson > gpurun_out/r6n_pool_scale_np.json 2> gpurun_out/r6n_pool_scale_np.err; grep '^pipeline' gpurun_out/r6n_pool_scale_np.err | cut -c1-260
timeout 300 python tools/probes/page_profile.py 96 2>&1 | sed -n 3,4p
