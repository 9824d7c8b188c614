#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_page_dispatch_diet.py tests/test_gpu_composite.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python tools/probes/page_profile.py 96 2>&1 | sed -n 3,12p | cut -c1-150
timeout 300 python tools/pool_scale.py --workers 1,8 --seconds 4 --modes pipeline 2>&1 | grep '^pipeline' | cut -c1-300
