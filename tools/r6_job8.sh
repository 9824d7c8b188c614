#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 300 python tools/probes/page_profile.py 96 > gpurun_out/r6c_page_profile.txt 2>&1; head -45 gpurun_out/r6c_page_profile.txt | cut -c1-150
VKX_LAYERS_MAPPED=0 timeout 300 python tools/probes/page_profile.py 96 2>&1 | sed -n 3,4p
