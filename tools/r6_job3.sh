#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
./tools/pool_trace.sh r6b 8 --no-poisson
./tools/pool_trace.sh r6b 1 --no-poisson
./tools/pool_trace.sh r6bp 8
./tools/pool_trace.sh r6b 4 --no-poisson
cat gpurun_out/r6b_pool_trace_w8.json | head -60
