#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_page_dispatch_diet.py tests/test_gpu_bench_c4.py tests/test_gpu_composite.py -x -q -m gpu > gpurun_out/r6f_tests.log 2>&1; echo "tests rc=$?"; tail -30 gpurun_out/r6f_tests.log
